#!/bin/bash
# round 3, int8 residual path: bench lines of the configs, the reference's run size, and a kernel trace of one cfg3 step
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for c in cfg2 cfg3 cfg5; do
  python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3i_bench_$c.json 2> gpurun_out/r3i_bench_$c.err || { tail -3 gpurun_out/r3i_bench_$c.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/r3i_bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['stages_ms'])"
done
python3 scripts/forest_scale.py > gpurun_out/r3i_forest_scale.json 2> gpurun_out/r3i_forest.err; cat gpurun_out/r3i_forest_scale.json
cd /tmp
rm -rf $R/gpurun_out/r3i_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3i_trace -o t -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3i_trace_bench.json 2> $R/gpurun_out/r3i_trace.log || { tail -5 $R/gpurun_out/r3i_trace.log; exit 1; }
python3 $R/scripts/trace_dump.py $(find $R/gpurun_out/r3i_trace -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r3i_timeline_cfg3.csv
cp $(find $R/gpurun_out/r3i_trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3i_cfg3_kernel_stats.csv
find $R/gpurun_out/r3i_trace -name "*.csv" -size +4M -delete
