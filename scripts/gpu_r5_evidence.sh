#!/bin/bash
# round-5 evidence of the current tree: default bench line, rocprofv3 kernel stats + timeline of cfg3, PMC traffic passes.
# usage: gpu_r5_evidence.sh  (writes gpurun_out/r5_*; copy what is to be judged into profiles/)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r5_bench_cfg3_default.json 2> gpurun_out/r5_bench_cfg3_default.err
echo "bench done: $(python3 -c "import json;d=json.loads(open('gpurun_out/r5_bench_cfg3_default.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['stages_ms'])")"
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_stats -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r5_bench_cfg3_under_rocprof_stats.json 2> $R/gpurun_out/r5_stats.err
F=$(find $R/gpurun_out/r5_stats -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/r5_cfg3_kernel_stats.csv
echo "stats done"
cd $R
bash scripts/gpu_trace_step.sh r5_cfg3 > gpurun_out/r5_cfg3_kernel_totals.txt 2>&1
rm -rf gpurun_out/r5_stats
bash scripts/gpu_pmc.sh r5 cfg3 > gpurun_out/r5_pmc.log 2>&1
cp gpurun_out/pmc_r5_traffic.json gpurun_out/pmc_traffic_cfg3.json
rm -rf gpurun_out/pmc_r5_FETCH_SIZE gpurun_out/pmc_r5_WRITE_SIZE
tail -5 gpurun_out/r5_pmc.log | cut -c1-300
