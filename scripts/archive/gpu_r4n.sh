#!/bin/bash
# round 4: the row-sharded float32-exchange layout on the GPU (gloo ranks sharing the one GPU) + rehearsal bench lines
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider --timeout=600 -x -k "row_sharded or int8_residual or active or two_rank_fit_matches" > gpurun_out/r4n_pytest.log 2>&1
echo "pytest exit=$?"; tail -6 gpurun_out/r4n_pytest.log
NNGP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --config cfg2 --mode shard32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4_rehearsal_shard32_cfg2_2ranks_gloo.json 2> gpurun_out/r4n_s32.err || tail -5 gpurun_out/r4n_s32.err
cat gpurun_out/r4_rehearsal_shard32_cfg2_2ranks_gloo.json | cut -c1-1500
NNGP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --config cfg2 --mode shard --steps 2 --warmup 1 --no-cpu-baseline --no-compare > gpurun_out/r4_rehearsal_shard_cfg2_2ranks_gloo.json 2> gpurun_out/r4n_s.err || tail -5 gpurun_out/r4n_s.err
python - <<'PY'
import json
for f in ("r4_rehearsal_shard32_cfg2_2ranks_gloo", "r4_rehearsal_shard_cfg2_2ranks_gloo"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["shard"], d["fit_info"]["alpha_l2"])
    except Exception as e:
        print(f, "failed", e)
PY
