#!/bin/bash
# kernel trace of one cfg3 bench step -> compact timeline CSV (gpurun_out/timeline_<tag>.csv)
TAG=${1:-t}
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -o t -- python3 $R/bench.py --config ${CFG:-cfg3} --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trace_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/trace_$TAG.log; exit 1; }
cd $R
T=$(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_dump.py $T gpurun_out/timeline_$TAG.csv
rm -rf gpurun_out/trace_$TAG
