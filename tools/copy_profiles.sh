#!/bin/bash
# Copies what `tools/final_round_run.sh <tag>` left under gpurun_out/ into the tracked profiles/<name>_* files.  Usage: tools/copy_profiles.sh r6b r6
SRC=${1:-r6}; DST=${2:-$SRC}
cd "$(dirname "$0")/.."
F=gpurun_out/final_$SRC; P=gpurun_out/prof_$SRC
cp $F/bench.json profiles/${DST}_bench.json
for n in config3 config3_eager config4 ddp_forced ddp_forced_config3 ddp_forced_config3_eager; do cp $F/bench_$n.json profiles/${DST}_bench_$n.json; done
cp $F/config3_busy.txt profiles/${DST}_config3_graph_busy.txt
cp $F/dv_time.txt profiles/${DST}_dv_time.txt
for n in dv_value_stationary self_split_time order_time winattn_time; do cp $F/$n.txt profiles/${DST}_$n.txt 2>/dev/null; done
for i in 1 2 3; do tail -1 $F/gputest_$i.log; done > profiles/${DST}_gputest_three_runs.txt 2>/dev/null
cp $F/library_roofline.txt profiles/${DST}_library_roofline.txt
cp $F/aten_call_sites.txt profiles/${DST}_aten_call_sites.txt
cp $F/aten_call_sites_config3.txt profiles/${DST}_aten_call_sites_config3.txt
cp $F/parity_e2e.json profiles/${DST}_parity_e2e.json
cp $P/kernel_stats.csv profiles/${DST}_bench_kernel_stats.csv
cp $P/domain_stats.csv profiles/${DST}_bench_domain_stats.csv
cp $P/stats_bench.json profiles/${DST}_bench_under_rocprof.json
cp $P/mfma_counters.json profiles/${DST}_mfma_counters.json
cp $P/pmc_traffic.json profiles/pmc_traffic.json
tail -3 $F/gputest.log
