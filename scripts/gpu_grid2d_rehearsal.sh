#!/bin/bash
# bench.py --mode grid2d with 4 ranks sharing the one GPU (gloo): end-to-end rehearsal of the 2-D block-cyclic layout on cfg2
export NNGP_DIST_BACKEND=gloo TMPDIR=/tmp
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus 4 --steps 1 --warmup 1 --config ${1:-cfg2} --mode grid2d > gpurun_out/rehearsal_grid2d.json 2> gpurun_out/rehearsal_grid2d.err
echo "exit=$?"; tail -2 gpurun_out/rehearsal_grid2d.err; tail -1 gpurun_out/rehearsal_grid2d.json | cut -c1-1200
