#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
NNGP_DEBUG=8=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_g -o t -- python3 $R/bench.py --config cfg3 --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/trace_g.log 2>&1
tail -1 $R/gpurun_out/trace_g.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'])"
python3 $R/scripts/r4_regions.py $(find $R/gpurun_out/trace_g -name "*kernel_trace.csv" | head -1)
