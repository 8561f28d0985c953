#!/bin/bash
# Same-session A/B of the drain work order (mode bit 6, msda_order_k): bench.py per-kernel timing + FETCH_SIZE of the drain, bin order
# (GE_MSDA_MODE=61) vs grouped by query range (default 125).     gpurun -- 'bash tools/ubench/ab_msda_order.sh'
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-fp32 --no-h2d --steps 20 --warmup 5"
python $REPO/bench.py $ARGS > /dev/null 2>&1          # warm caches
for mode in 61 125 61 125; do
  GE_MSDA_MODE=$mode python $REPO/bench.py $ARGS 2>/dev/null > $OUT/bench_$mode.json
  python - $OUT/bench_$mode.json $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
ks = {k['name']: k for k in d['kernels']}
pick = [k for k in d['kernels'] if 'drain' in k['name'] or 'bwd_value' in k['name'] or 'msda_bwd_raw' in k['name']]
print(f"mode {sys.argv[2]}: {d['ms_per_step']} ms/step; " + '; '.join(f"{k['name']} {k['avg_us']} us" for k in pick))
PY
done
for mode in 61 125; do
  GE_MSDA_MODE=$mode rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$mode -- python $REPO/bench.py --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing --steps 1 --warmup 1 > /dev/null 2> $OUT/fetch_$mode.err
  GE_MSDA_MODE=$mode rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$mode -- python $REPO/bench.py --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing --steps 1 --warmup 1 > /dev/null 2> $OUT/write_$mode.err
  python - $OUT $mode <<'PY'
import csv, glob, sys, collections
out, mode = sys.argv[1], sys.argv[2]
for what in ('fetch', 'write'):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{what}_{mode}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'drain' in k or 'hist' in k or 'order' in k:
                agg[k].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f'mode {mode} {what.upper()}_SIZE raw counter (uncorrected: see tools/pmc_traffic.py) {k[:60]:60s} n={len(v)} mean={sum(v)/len(v):.4g}')
PY
  rm -rf $OUT/fetch_$mode $OUT/write_$mode
done
