#!/bin/bash
# kernel trace of one bench step under NNGP_DEBUG variants; prints the Cholesky timeline summary
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for V in "$@"; do
  i=$((i+1))
  export NNGP_DEBUG=$V
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$i -o t -- python3 $R/bench.py --config cfg3 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trace_$i.log 2>&1 || exit 1
  echo "== NNGP_DEBUG=$V"
  python3 $R/scripts/trace_timeline.py $(find $R/gpurun_out/trace_$i -name "*kernel_trace.csv" | head -1)
done
