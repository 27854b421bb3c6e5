#!/bin/bash
# Round-2 GPU session G: the profile set of the round (copied to profiles/r2_* afterwards).
set -o pipefail
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out/r2
O=gpurun_out/r2
export TMPDIR=/tmp
if [ "${PART:-A}" = "A" ]; then
NNGP_FULL_ORACLE=${FULL:-0} timeout -k 10 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout=900 > $O/pytest_gpu.log 2>&1
echo "pytest exit=$?"; grep -a "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.log | cut -c1-300 | tail -6
grep -a "CONFIG_CHECK" $O/pytest_gpu.log > $O/config_checks.txt
for C in cfg1 cfg2 cfg4 cfg5; do
  timeout -k 10 400 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$C.json 2> $O/bench_$C.err || tail -3 $O/bench_$C.err
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || tail -3 $O/bench_cfg3.err
for C in cfg1 cfg2 cfg3 cfg4 cfg5; do python -c "
import json
d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]); print('$C', d['ms_per_step'], d['stages_ms'], d['roofline']['frac'], d['roofline_posterior']['frac'], d['roofline_k1']['frac'])"; done
fi
if [ "${PART:-A}" = "B" ]; then
# rocprofv3 kernel stats + timeline of the same bench command
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof.log 2>&1; echo "rocprof exit=$?"; cd $R
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 scripts/trace_dump.py $T $O/timeline_cfg3.csv; cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/cfg3_kernel_stats.csv; rm -rf $O/prof
# HBM-side traffic (separate --pmc passes, kernel trace only)
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc_$C -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/pmc_$C.log 2>&1; echo "$C exit=$?"
done
cd $R
python3 scripts/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE cfg3 > $O/pmc_traffic_cfg3.json && python -c "
import json; d=json.load(open('$O/pmc_traffic_cfg3.json')); print({k:(round(v['fetch_bytes']/1e9,2), round(v['write_bytes']/1e9,2)) for k,v in d['kernels'].items()}, d['cholesky_bytes']/1e9)"
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
timeout -k 10 200 python scripts/forest_scale.py > $O/forest_scale.json 2>/dev/null; cat $O/forest_scale.json
timeout -k 10 300 python scripts/serving_latency.py > $O/serving_latency.json 2>/dev/null; head -c 600 $O/serving_latency.json; echo
timeout -k 10 300 python scripts/estimator_latency.py > $O/estimator_latency.json 2>/dev/null; head -c 400 $O/estimator_latency.json; echo
timeout -k 10 300 python scripts/append_bench.py > $O/append_bench.json 2>/dev/null; head -c 400 $O/append_bench.json; echo
timeout -k 10 300 python scripts/var_study.py > $O/var_study.json 2>/dev/null
timeout -k 10 200 python scripts/k1_variants.py 2>/dev/null | grep KC > $O/k1_variants.txt
timeout -k 10 200 python scripts/k1_study.py > $O/k1_study.json 2>/dev/null
make -C scripts/micro -s && ./scripts/micro/f64_pipes > $O/micro_f64_pipes.txt 2>&1; ./scripts/micro/f64_seed_accuracy > $O/micro_f64_seed_accuracy.txt 2>&1
bash scripts/gpu_multirank_rehearsal.sh cfg2 2>&1 | tail -5; for g in 1 2 4; do cp gpurun_out/rehearsal_g$g.json $O/rehearsal_shard_cfg2_${g}ranks_gloo.json; done
ab() {
  NNGP_DEBUG=$1 timeout -k 10 300 python bench.py --config $2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug=$1 $2', d['ms_per_step'], d['stages_ms']['cholesky'], d['fit_info']['cg_iters'])"
}
for r in 24 32 48; do ab "4=$r" cfg3; done
fi
