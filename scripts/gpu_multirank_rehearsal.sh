#!/bin/bash
# Rehearse bench.py --gpus 2 / 4 on a ONE-GPU box: ranks share cuda:0, collectives via gloo (host staging).
# Checks the sharded build + all-gather + replicated factor path end to end on the real kernels (small config).
export NNGP_DIST_BACKEND=gloo TMPDIR=/tmp
for G in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29500+G)) \
     bench.py --gpus $G --steps 2 --warmup 1 --config cfg2 > gpurun_out/rehearsal_g$G.json 2> gpurun_out/rehearsal_g$G.err
  echo "gpus=$G exit=$?"; cat gpurun_out/rehearsal_g$G.json | cut -c1-900; tail -3 gpurun_out/rehearsal_g$G.err
done
