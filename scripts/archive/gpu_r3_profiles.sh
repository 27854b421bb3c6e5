#!/bin/bash
# round 3: the profiles the bench line's roofline object refers to (run on the GPU box; summaries go to gpurun_out/, copy to profiles/)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp
# 1. per-kernel time of the bench command itself
rm -rf $R/gpurun_out/r3_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_stats -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3_stats_bench.json 2> $R/gpurun_out/r3_stats.log || { tail -5 $R/gpurun_out/r3_stats.log; exit 1; }
cp $(find $R/gpurun_out/r3_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3_cfg3_kernel_stats.csv
find $R/gpurun_out/r3_stats -name "*.csv" -size +4M -delete
cd $R
# 2. fabric traffic per kernel (FETCH_SIZE x2 + WRITE_SIZE, separate passes)
bash scripts/gpu_pmc.sh r3 cfg3 > gpurun_out/r3_pmc_traffic.log 2>&1; tail -3 gpurun_out/r3_pmc_traffic.log
# 3. SQ / TCP / TCC counters of the dominant kernel on its two shapes
bash scripts/gpu_h3_pmc.sh > gpurun_out/r3_h3_pmc_raw.txt 2>&1; tail -40 gpurun_out/r3_h3_pmc_raw.txt
