#!/bin/bash
# round 3 (int8 residual): bench lines of every config + the reference's run size + the full-size config checks (with their measured errors)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for c in cfg1 cfg2 cfg4 cfg5; do
  python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3i_bench_$c.json 2> gpurun_out/r3i_bench_$c.err || { tail -3 gpurun_out/r3i_bench_$c.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/r3i_bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['stages_ms'], d['roofline']['frac'])"
done
python3 scripts/forest_scale.py > gpurun_out/r3i_forest_scale.json 2> gpurun_out/r3i_forest.err; cat gpurun_out/r3i_forest_scale.json
python -m pytest tests/test_gpu_configs.py -q -m gpu -s > gpurun_out/r3i_config_checks.txt 2>&1; grep -c CONFIG_CHECK gpurun_out/r3i_config_checks.txt; tail -2 gpurun_out/r3i_config_checks.txt
