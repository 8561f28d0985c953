#!/bin/bash
# rocprofv3 PMC pass over ONE bench step: per kernel, how busy are its waves?  (SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_LDS count quad-cycles
# summed over waves.)  Output: gpurun_out/pmc_step/summary.txt — kernels by total wave-cycles with the share of them spent issuing / waiting.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_step
mkdir -p $OUT
ARGS="--no-cpu-baseline --no-fp32 --no-kernel-timing --no-h2d --steps 1 --warmup 2"
python $REPO/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-fp32 --no-kernel-timing --no-h2d > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/raw -- python $REPO/bench.py $ARGS > $OUT/run.log 2>&1
REPO=$REPO python - <<'PY'
import csv, glob, os, collections
out = os.environ['REPO'] + '/gpurun_out/pmc_step'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(out + '/raw/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES':
            n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))
with open(out + '/summary.txt', 'w') as fh:
    fh.write(f'{"kernel":70s} launches  wave-cycles   active%  wait%  lds-wait%  instr/wave  valu/wave lds/wave\n')
    for k, d in rows[:60]:
        wc = max(d.get('SQ_WAVE_CYCLES', 0), 1); w = max(d.get('SQ_WAVES', 0), 1)
        fh.write(f'{k:70s} {n[k]:8d} {wc:12.3g} {100 * d.get("SQ_ACTIVE_INST_ANY", 0) / wc:8.1f} {100 * d.get("SQ_WAIT_ANY", 0) / wc:6.1f} {100 * d.get("SQ_WAIT_INST_LDS", 0) / wc:9.1f} '
                 f'{(d.get("SQ_INSTS_VALU", 0) + d.get("SQ_INSTS_LDS", 0)) / w:10.0f} {d.get("SQ_INSTS_VALU", 0) / w:9.0f} {d.get("SQ_INSTS_LDS", 0) / w:7.0f}\n')
print(open(out + '/summary.txt').read())
PY
rm -rf $OUT/raw
