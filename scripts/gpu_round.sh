#!/bin/bash
# One GPU-box session: parity tests, bench lines, rocprof kernel stats.  Usage: scripts/gpu_round.sh <tag> [skip_tests]
set -o pipefail
TAG=${1:-r1}
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ "$2" != "skip_tests" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 > gpurun_out/pytest_gpu_$TAG.log 2>&1
  echo "pytest exit=$?" | tee -a gpurun_out/pytest_gpu_$TAG.log
  tail -3 gpurun_out/pytest_gpu_$TAG.log
fi
timeout -k 10 300 python bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg2_$TAG.json 2> gpurun_out/bench_cfg2_$TAG.err || { echo "bench cfg2 failed"; tail -5 gpurun_out/bench_cfg2_$TAG.err; }
cat gpurun_out/bench_cfg2_$TAG.json
timeout -k 10 600 python bench.py --config cfg3 --steps 3 --warmup 1 > gpurun_out/bench_cfg3_$TAG.json 2> gpurun_out/bench_cfg3_$TAG.err || { echo "bench cfg3 failed"; tail -5 gpurun_out/bench_cfg3_$TAG.err; }
cat gpurun_out/bench_cfg3_$TAG.json
timeout -k 10 300 python scripts/microbench.py > gpurun_out/microbench_$TAG.json 2> gpurun_out/microbench_$TAG.err; cat gpurun_out/microbench_$TAG.json | tr -d "\n" | cut -c1-3000; echo
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit=$?"
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_$TAG -name "*stats*" | head; find gpurun_out/prof_$TAG -name "*kernel_stats*" -exec head -25 {} \;
# keep the merge-back small: drop the per-dispatch trace, keep the stats
find gpurun_out/prof_$TAG -name "*kernel_trace*" -size +20M -delete
