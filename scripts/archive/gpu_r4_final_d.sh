#!/bin/bash
# round 4: after the last library change (solve timer) -- full suite, evidence B, then the bench lines with the refreshed PMC file
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_pytest_gpu.log
bash scripts/gpu_r4_evidence.sh 2>&1 | tail -6
cp gpurun_out/pmc_r4_traffic.json profiles/pmc_traffic_cfg3.json
for C in cfg1 cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 400 python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_bench_$C.json 2> gpurun_out/r4_bench_$C.err; echo "$C rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_$C.json').read().strip().splitlines()[-1]); print('$C', d['ms_per_step'], d['stages_ms'], d['roofline']['frac'], d.get('roofline_residual',{}).get('frac'), d.get('roofline_solves',{}).get('frac'), d.get('roofline_solves',{}).get('ms_per_step'))"
done
timeout -k 10 600 python bench.py > gpurun_out/r4_bench_cfg3_default_with_cpu_baseline.json 2> gpurun_out/r4_bench_default.err; echo "default rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_cfg3_default_with_cpu_baseline.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline_solves'])"
