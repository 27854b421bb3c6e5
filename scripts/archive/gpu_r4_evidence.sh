#!/bin/bash
# round 4: refresh of the evidence the judge asked for (serving / estimator / append latencies on the current tree, forest-size run,
# what the alpha CG costs beside the covariance), the kernel stats + timeline of the bench command, and the fabric traffic per kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python scripts/serving_latency.py > gpurun_out/r4_serving_latency.json 2> gpurun_out/r4_serving_latency.err; echo "serving rc=$?"
timeout -k 10 300 python scripts/estimator_latency.py > gpurun_out/r4_estimator_latency.json 2> gpurun_out/r4_estimator_latency.err; echo "estimator rc=$?"
timeout -k 10 300 python scripts/append_bench.py > gpurun_out/r4_append_bench.json 2> gpurun_out/r4_append_bench.err; echo "append rc=$?"
timeout -k 10 300 python scripts/forest_scale.py > gpurun_out/r4_forest_scale.json 2> gpurun_out/r4_forest_scale.err; echo "forest rc=$?"
timeout -k 10 300 python scripts/cov_alone.py > gpurun_out/r4_cov_alone.json 2> gpurun_out/r4_cov_alone.err; echo "cov_alone rc=$?"; cat gpurun_out/r4_cov_alone.json
cd /tmp
rm -rf $R/gpurun_out/r4_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_stats -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4_bench_cfg3_under_rocprof_stats.json 2> $R/gpurun_out/r4_stats.log || tail -5 $R/gpurun_out/r4_stats.log
cp $(find $R/gpurun_out/r4_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4_cfg3_kernel_stats.csv
python3 $R/scripts/trace_dump.py $(find $R/gpurun_out/r4_stats -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r4_timeline_cfg3.csv
find $R/gpurun_out/r4_stats -name "*.csv" -size +4M -delete
cd $R
bash scripts/gpu_pmc.sh r4 cfg3 > gpurun_out/r4_pmc_traffic.log 2>&1; tail -3 gpurun_out/r4_pmc_traffic.log
cp gpurun_out/pmc_r4_traffic.json gpurun_out/r4_pmc_traffic_cfg3.json
