#!/usr/bin/env python
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950) of `bench.py` into
profiles/pmc_traffic.json: HBM bytes per launch for every kernel of libgedepth_hip.so.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/pmc_fetch -- python bench.py --steps 1 --warmup 1 ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/pmc_write -- python bench.py --steps 1 --warmup 1 ...
    python tools/pmc_traffic.py --fetch out/pmc_fetch --write out/pmc_write --adamw-elems <n_params> --out profiles/pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): the counters are in KiB of memory-side L2 requests; on gfx950 their
absolute scale depends on the access width, so both are calibrated on a kernel of this same run whose traffic is known
exactly and exceeds the 256 MiB Infinity Cache: the fused AdamW step (reads p, g, m, v + a 1-byte decay mask = 17 bytes per
parameter; writes p, m, v = 12 bytes per parameter, or 14 for the `adamw_k<true>` variant, which also writes the bf16 shadow copy
of the parameters — round 2 calibrated WRITE_SIZE with 12 on that variant, so its write figures were 14 % low).  The calibration
factors are stored next to the numbers, with a cross-check on `sumsq_k` (reads exactly 4 bytes per arena element).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

OURS = ('msda_', 'window_attn', 'bilinear_', 'bias_act_', 'tokens_from_map', 'map_from_tokens', 'ground_', 'depth_fuse',
        'silog_', 'sumsq_k', 'adamw_k', 'pe_channels', 'slope_class', 'upcat_', 'upsum_', 'bias_gelu', 'bn_', 'colsum_k', 'concat_rows',
        'slice_rows', 'add_rows', 'layernorm_', 'residual_', 'scale_rows', 'conv3x3_', 'mw_prepare', 'aug_')


def short(name):
    name = re.sub(r'^void\s+', '', name.strip())
    return name.split('(')[0].strip()


def collect(directory, counter):
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no *counter_collection.csv under {directory}')
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                k = short(row['Kernel_Name'])
                if not k.startswith(OURS):
                    continue
                acc[k][0] += float(row['Counter_Value'])
                acc[k][1] += 1
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fetch', required=True)
    ap.add_argument('--write', required=True)
    ap.add_argument('--adamw-elems', type=int, required=True, help='parameters updated by one ge_adamw_step launch')
    ap.add_argument('--out', default='profiles/pmc_traffic.json')
    ap.add_argument('--note', default='')
    a = ap.parse_args()
    fetch, write = collect(a.fetch, 'FETCH_SIZE'), collect(a.write, 'WRITE_SIZE')
    cal = {}
    for name, acc, per_elem in (('FETCH_SIZE', fetch, 17), ('WRITE_SIZE', write, 12)):
        key = next((k for k in acc if k.startswith('adamw_k')), None)
        if key is None:
            raise SystemExit('calibration kernel adamw_k not in the trace')
        if name == 'WRITE_SIZE' and 'true' in key:                      # adamw_k<true>: + the 2-byte bf16 shadow of every parameter
            per_elem = 14
        reported = acc[key][0] / acc[key][1] * 1024.0
        cal[name] = dict(kernel=key, known_bytes=per_elem * a.adamw_elems, reported_bytes=reported,
                         factor=per_elem * a.adamw_elems / reported)
    sk = next((k for k in fetch if k.startswith('sumsq_k')), None)
    if sk is not None:                                                    # independent check of the FETCH factor
        got = fetch[sk][0] / fetch[sk][1] * 1024.0 * cal['FETCH_SIZE']['factor']
        cal['check_sumsq_k'] = dict(fetch_bytes=round(got), bytes_per_adamw_elem=got / a.adamw_elems)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [0.0, 0])
        w = write.get(k, [0.0, 0])
        fb = f[0] / f[1] * 1024.0 * cal['FETCH_SIZE']['factor'] if f[1] else 0.0
        wb = w[0] / w[1] * 1024.0 * cal['WRITE_SIZE']['factor'] if w[1] else 0.0
        kernels[k] = dict(launches=max(f[1], w[1]), fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb),
                          hbm_bytes_per_launch=round(fb + wb),
                          uncalibrated_bytes_per_launch=round((f[0] / f[1] if f[1] else 0.0) * 1024 + (w[0] / w[1] if w[1] else 0.0) * 1024))
    out = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py', note=a.note,
               calibration=cal, kernels=kernels)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(out, fh, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:12]:
        print(f"{k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == '__main__':
    main()
