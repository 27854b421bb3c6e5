#!/bin/bash
# HBM traffic of one bench step from the TCC counters (separate --pmc passes, kernel-trace only, as the guide prescribes).
TAG=${1:-r1}
CFG=${2:-cfg3}
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C -o $CFG -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C.log 2>&1
  echo "$C exit=$?"
done
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/pmc_${TAG}_FETCH_SIZE/ | head
python3 scripts/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE $CFG > gpurun_out/pmc_${TAG}_traffic.json && cat gpurun_out/pmc_${TAG}_traffic.json
# keep only the per-kernel aggregate (the per-dispatch CSVs are large)
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +8M -delete
