#!/bin/bash
# Round-2 GPU session C: GPU suite, cfg3 bench line with the CPU baseline, rocprof kernel stats + per-kernel timeline of one step.
set -o pipefail
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu_r2c.log 2>&1
echo "pytest exit=$?"; grep -a "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu_r2c.log | cut -c1-300 | tail -12
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_cfg3_r2c.json 2> gpurun_out/bench_cfg3_r2c.err || tail -5 gpurun_out/bench_cfg3_r2c.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_cfg3_r2c.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms']); print(json.dumps(d['cpu_baseline'])[:1500])"
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2c -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r2c.log 2>&1
echo "rocprof exit=$?"
cd $R
T=$(find gpurun_out/prof_r2c -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_dump.py $T gpurun_out/timeline_cfg3_r2c.csv
python3 scripts/trace_timeline.py $T | head -60
find gpurun_out/prof_r2c -name "*kernel_stats*" -exec head -30 {} \; | cut -c1-150
find gpurun_out/prof_r2c -name "*kernel_trace*" -size +20M -delete
