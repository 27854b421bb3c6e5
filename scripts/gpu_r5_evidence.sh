#!/bin/bash
# round-5 evidence of the current tree, in the order the bench line needs it: PMC traffic passes (their summary goes to profiles/ so that the
# bench can quote `traffic`), the default bench line, rocprofv3 kernel stats, the timeline of one step.
# usage: gpu_r5_evidence.sh  (writes gpurun_out/r5_*; copy what is to be judged into profiles/)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/gpu_pmc.sh r5 cfg3 > gpurun_out/r5_pmc.log 2>&1
cp gpurun_out/pmc_r5_traffic.json gpurun_out/pmc_traffic_cfg3.json
cp gpurun_out/pmc_r5_traffic.json profiles/pmc_traffic_cfg3.json
rm -rf gpurun_out/pmc_r5_FETCH_SIZE gpurun_out/pmc_r5_WRITE_SIZE
echo "pmc done: $(tail -c 300 gpurun_out/r5_pmc.log | tr '\n' ' ')"
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r5_bench_cfg3_default.json 2> gpurun_out/r5_bench_cfg3_default.err
echo "bench done: $(python3 -c "import json;d=json.loads(open('gpurun_out/r5_bench_cfg3_default.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['stages_ms'], 'traffic', d['roofline'].get('traffic'), d['roofline_solves'].get('traffic'))")"
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_stats -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r5_bench_cfg3_under_rocprof_stats.json 2> $R/gpurun_out/r5_stats.err
F=$(find $R/gpurun_out/r5_stats -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/r5_cfg3_kernel_stats.csv
echo "stats done"
cd $R
bash scripts/gpu_trace_step.sh r5_cfg3 > gpurun_out/r5_cfg3_kernel_totals.txt 2>&1
rm -rf gpurun_out/r5_stats
tail -8 gpurun_out/r5_cfg3_kernel_totals.txt
