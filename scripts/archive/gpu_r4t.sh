#!/bin/bash
# round 4: kernel timeline of the bench step on the current tree
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp
rm -rf $R/gpurun_out/r4t_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4t_stats -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4t_bench.json 2> $R/gpurun_out/r4t_stats.log || tail -5 $R/gpurun_out/r4t_stats.log
cp $(find $R/gpurun_out/r4t_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4t_kernel_stats.csv
python3 $R/scripts/trace_dump.py $(find $R/gpurun_out/r4t_stats -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r4t_timeline_cfg3.csv
find $R/gpurun_out/r4t_stats -name "*.csv" -size +4M -delete
