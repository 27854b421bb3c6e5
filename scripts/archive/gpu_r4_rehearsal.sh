#!/bin/bash
# round 4, final tree: bench.py --gpus 2 / 4 with ranks sharing one MI355X through gloo, in the three layouts (shard, shard32, replicate)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for MODE in shard shard32 replicate; do
  echo "=== mode $MODE"
  NNGP_DIST_MODE=$MODE bash scripts/gpu_multirank_rehearsal.sh cfg2 2>&1 | tail -6
  cp gpurun_out/rehearsal_g2.json gpurun_out/r4_rehearsal_${MODE}_cfg2_2ranks_gloo.json
  cp gpurun_out/rehearsal_g4.json gpurun_out/r4_rehearsal_${MODE}_cfg2_4ranks_gloo.json
done
