#!/bin/bash
# fabric-side traffic of one cfg3 step (separate --pmc passes, kernel trace only) -> gpurun_out/r2/pmc_traffic_cfg3.json
cd "$(dirname "$0")/.."; R=$(pwd); O=gpurun_out/r2; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc_$C -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/pmc_$C.log 2>&1; echo "$C exit=$?"
done
cd $R
python3 scripts/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE cfg3 > $O/pmc_traffic_cfg3.json && python -c "
import json; d=json.load(open('$O/pmc_traffic_cfg3.json')); print({k:(round(v['fetch_bytes']/1e9,2), round(v['write_bytes']/1e9,2)) for k,v in d['kernels'].items()}, d['cholesky_bytes']/1e9)"
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
