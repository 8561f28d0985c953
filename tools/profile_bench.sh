#!/bin/bash
# Evidence for bench.py's `roofline` object: rocprofv3 kernel stats + HBM traffic (PMC) of the same command.
#   gpurun -- 'bash tools/profile_bench.sh r2'      -> gpurun_out/prof_r2/{kernel_stats.csv, domain_stats.csv, pmc_traffic.json}
# Passes are separate (gpurun refuses --pmc combined with other trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
TAG=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-fp32 --no-kernel-timing --no-h2d ${BENCH_EXTRA:-}"      # e.g. BENCH_EXTRA="--layout nchw"
# warm MIOpen / TunableOp caches in this session (a cold MIOpen under rocprofv3 falls back to naive_conv kernels)
python $REPO/bench.py --steps 3 --warmup 3 $ARGS > $OUT/warm.json 2> $OUT/warm.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --steps 20 --warmup 5 $ARGS > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py --steps 1 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py --steps 1 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python $REPO/bench.py --steps 1 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_mfma.err
python $REPO/tools/pmc_mfma.py --dir $OUT/pmc_mfma --out $OUT/mfma_counters.json > $OUT/mfma_top.txt 2>&1
NP=$(python -c "import json;print(json.load(open('$OUT/warm.json'))['config']['params'])")
python $REPO/tools/pmc_traffic.py --fetch $OUT/pmc_fetch --write $OUT/pmc_write --adamw-elems $NP --out $OUT/pmc_traffic.json --note "round $TAG" > $OUT/pmc_top.txt 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
cp $(find $OUT/stats -name '*domain_stats.csv' | head -1) $OUT/domain_stats.csv 2>/dev/null
rm -rf $OUT/stats/*/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma      # keep the merge under the 64 MiB limit
head -30 $OUT/kernel_stats.csv; cat $OUT/pmc_top.txt; cat $OUT/mfma_top.txt; cat $OUT/stats_bench.json | head -c 600
