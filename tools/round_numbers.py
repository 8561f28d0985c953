#!/usr/bin/env python
"""Prints the numbers DESIGN.md §6 / README quote from the tracked profiles of one round:  python tools/round_numbers.py r6"""
import csv
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r6'
P = 'profiles/' + tag + '_'
d = json.load(open(P + 'bench.json'))
print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms | h2d', d['with_h2d']['value'], '| fp32', d['fp32']['value'], d['fp32']['ms_per_step'])
r = d['roofline']
print('roofline', r['kernel'], r['avg_us'], 'us', r['achieved'], 'GB/s frac', r['frac'], 'traffic', r['traffic'])
for f in ['config3', 'config3_eager', 'config4', 'ddp_forced', 'ddp_forced_config3', 'ddp_forced_config3_eager']:
    try:
        x = json.load(open(P + 'bench_' + f + '.json'))
        extra = ''
        if x.get('ddp', {}).get('trace'):
            t = x['ddp']['trace']
            extra = f" buckets {len(t)} first launch {t[0]['launch_ms']:.1f} ms last {t[-1]['MB']} MB at {t[-1]['launch_ms']:.1f} / done {t[-1]['done_ms']:.1f}"
        print(f, x['value'], x['ms_per_step'], x['config'].get('hip_graph'), extra)
    except FileNotFoundError:
        print(f, 'missing')
rows = list(csv.DictReader(open(P + 'bench_kernel_stats.csv')))
n = [int(r_['Calls']) for r_ in rows if 'adamw_k' in r_['Name']][0]          # one optimizer launch per step (since round 6 msda_mm_bwd_lw_k runs twice per step)
drain = [r_ for r_ in rows if 'msda_drain' in r_['Name']][0]
print('steps', n, 'kernel ms/step', round(sum(float(r_['TotalDurationNs']) for r_ in rows) / 1e6 / n, 2), 'launches/step', round(sum(int(r_['Calls']) for r_ in rows) / n))
print('drain calls', drain['Calls'], 'avg us', float(drain['AverageNs']) / 1e3)


def grp(pred):
    return round(sum(float(r_['TotalDurationNs']) for r_ in rows if pred(r_['Name'])) / 1e6 / n, 2)


lib = lambda s: 'igemm' in s or 'ck::tensor_operation' in s or '_ZN2ck16tensor' in s or 'SubTensor' in s
print(dict(msda=grp(lambda s: 'msda' in s), lib_gemm=grp(lambda s: s.startswith('Cijk') or s.startswith('Custom_Cijk') or 'kernel_batched_gemm' in s),
           conv3x3=grp(lambda s: s.startswith('conv3x3')), conv1x1_bn=grp(lambda s: 'conv1x1' in s), gemm_nt=grp(lambda s: 'gemm_nt_k' in s),
           bn=grp(lambda s: 'bn_' in s and 'conv1x1' not in s), window=grp(lambda s: 'window_attn' in s), ln=grp(lambda s: 'layernorm' in s),
           miopen_ck=grp(lib), aten=grp(lambda s: 'at::native' in s)))
for k in ('msda_mm_fwd_k', 'msda_mm_bwd_lw_k', 'msda_fwd_win_k', 'msda_bwd_lw_k<', 'msda_drain', 'msda_hist_raw_k<true>', 'msda_hist_raw_k<false>', 'conv1x1_bn_act_k',
          'conv1x1_bn_mask_k', 'conv1x1_wgrad_k', 'conv1x1_bn_dgrad_k'):
    print('  ', k, grp(lambda s, k=k: k in s))
