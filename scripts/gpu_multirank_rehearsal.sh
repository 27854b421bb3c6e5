#!/bin/bash
# Rehearse bench.py --gpus 2 / 4 on a ONE-GPU box: ranks share cuda:0, collectives via gloo (host staging).
# Checks the sharded build + all-gather + block-cyclic Cholesky path end to end on the real kernels, and that
# alpha agrees with the single-rank run (alpha_l2 printed in fit_info).
export NNGP_DIST_BACKEND=gloo TMPDIR=/tmp
export NNGP_DIST_MODE=${NNGP_DIST_MODE:-shard}   # the sharded layout is what needs rehearsing; 'replicate' has no collective
CFG=${1:-cfg2}
timeout -k 10 300 python bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/rehearsal_g1.json 2> gpurun_out/rehearsal_g1.err
for G in 2 4; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29500+G)) \
     bench.py --gpus $G --steps 1 --warmup 1 --config $CFG > gpurun_out/rehearsal_g$G.json 2> gpurun_out/rehearsal_g$G.err
  echo "gpus=$G exit=$?"; tail -2 gpurun_out/rehearsal_g$G.err
done
python - <<'PY'
import json
ref = None
for g in (1, 2, 4):
    d = json.loads(open("gpurun_out/rehearsal_g%d.json" % g).read().strip().splitlines()[-1])
    print(g, d["fit_info"], {k: round(v, 2) for k, v in d.get("stages_ms", {}).items()})
    if ref is None:
        ref = d["fit_info"]["alpha_l2"]
    assert abs(d["fit_info"]["alpha_l2"] - ref) <= 1e-9 * abs(ref), "alpha differs between rank counts"
    assert d["fit_info"].get("clamped_pivots", 0) == 0 and d["fit_info"]["rel_residual"] < 1e-9
print("multi-rank rehearsal OK")
PY
