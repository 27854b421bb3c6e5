#!/bin/bash
# round 4, final tree: the default bench command and cfg3 once more, now that the PMC traffic file carries this tree's fingerprint
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_bench_cfg3.json 2> gpurun_out/r4_bench_cfg3.err; echo "cfg3 rc=$?"
timeout -k 10 600 python bench.py > gpurun_out/r4_bench_cfg3_default_with_cpu_baseline.json 2> gpurun_out/r4_bench_default.err; echo "default rc=$?"
python3 -c "
import json
for f in ('gpurun_out/r4_bench_cfg3.json','gpurun_out/r4_bench_cfg3_default_with_cpu_baseline.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_over_algorithmic'), d['roofline_residual']['frac'], d['roofline_residual']['traffic'])"
