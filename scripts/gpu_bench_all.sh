#!/bin/bash
# the bench line of every BASELINE config on one GPU + the forest-sized run: gpu_bench_all.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=$1
for C in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python3 $R/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_$C.json 2> $R/gpurun_out/${TAG}_bench_$C.err
  python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/${TAG}_bench_$C.json").read().strip().splitlines()[-1])
s = d.get("roofline_solves", {})
print("$C", "step %.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in d["stages_ms"].items() if v > 0.05}, "solves", s.get("ms_per_step"), s.get("frac"), "roofline", d["roofline"]["frac"], "cg", d["fit_info"]["cg_iters"], flush=True)
PY
done
python3 $R/scripts/forest_scale.py > $R/gpurun_out/${TAG}_forest_scale.json 2>/dev/null; cat $R/gpurun_out/${TAG}_forest_scale.json
