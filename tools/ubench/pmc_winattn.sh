#!/bin/bash
# rocprofv3 PMC passes over the hand-written 3x3 convolution kernels (tools/ubench/winattn_time.py): where do conv3x3_nhwc_k / conv3x3_wgrad_k wait?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_winattn
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python $REPO/tools/ubench/winattn_time.py > $OUT/$tag.log 2>&1
done
REPO=$REPO python - <<'PY'
import csv, glob, os, collections
out = os.environ['REPO'] + '/gpurun_out/pmc_winattn'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'window_attn' in k:
            agg[k + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k, d in sorted(agg.items()):
        fh.write(k + '\n')
        for c, v in sorted(d.items()):
            fh.write(f'   {c:40s} n={len(v):4d} mean={sum(v)/len(v):.4g}\n')
print(open(out + '/summary.txt').read()[:6000])
PY
