#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one bench step per kernel under environment switches: gpu_pmc_ab.sh "<env>" ["<env>" ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for V in "$@"; do
  i=$((i+1))
  for C in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    export $V
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcab_${i}_$C -o cfg3 -- python3 $R/bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcab_${i}_$C.log 2>&1
    cd $R
  done
  python3 scripts/pmc_traffic.py gpurun_out/pmcab_${i}_FETCH_SIZE gpurun_out/pmcab_${i}_WRITE_SIZE cfg3 > gpurun_out/pmcab_${i}.json
  echo "$V: $(python3 -c "
import json
d = json.load(open('gpurun_out/pmcab_${i}.json'))
k = d.get('kernels', d)
for name in ('k_trsm_tickets', 'k_gemm_nt_i8s'):
    if name in k: print(name, json.dumps(k[name]))
")"
  rm -rf gpurun_out/pmcab_${i}_FETCH_SIZE gpurun_out/pmcab_${i}_WRITE_SIZE
done
