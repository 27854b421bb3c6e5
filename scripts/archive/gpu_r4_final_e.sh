#!/bin/bash
# the default bench command (20 steps after 3 warm-up), twice, on the final tree
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 600 python bench.py > gpurun_out/r4_bench_default_$i.json 2> gpurun_out/r4_bench_default_$i.err; echo "default rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_default_$i.json').read().strip().splitlines()[-1]); print(d['steps'], d['warmup'], d['ms_per_step'], d['value'], d['stages_ms'], d['roofline']['frac'], d['roofline']['traffic_over_algorithmic'], d['roofline_residual']['frac'], d['roofline_solves']['frac'])"
done
