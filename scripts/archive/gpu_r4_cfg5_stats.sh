#!/bin/bash
# round 4: kernel stats of the NTK config (cfg5) -- where its 32 ms of posterior go
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp; rm -rf $R/gpurun_out/r4_cfg5_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_cfg5_stats -o cfg5 -- python3 $R/bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4_cfg5_under_rocprof.json 2> $R/gpurun_out/r4_cfg5_stats.log
cp $(find $R/gpurun_out/r4_cfg5_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4_cfg5_kernel_stats.csv
python3 $R/scripts/trace_dump.py $(find $R/gpurun_out/r4_cfg5_stats -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r4_timeline_cfg5.csv
find $R/gpurun_out/r4_cfg5_stats -name "*.csv" -size +4M -delete
head -30 $R/gpurun_out/r4_cfg5_kernel_stats.csv | cut -c1-150
