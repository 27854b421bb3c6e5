"""Cost of splitting the test rows into blocks (the premise of a block-pipelined posterior): predict(M) at cfg3's N for
M = 1024, 512, 256 test rows, same fit; per stage of the level-1 variance the float64 product scales with M, the three
blocked solves (32 latency-bound steps each) hardly do."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d = 32768, 128
x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(1024, d, seed=1)
model = GPModel(n, d, [1.0] * 4, [0.0] * 4, diag_reg=1e-3, m_cap=1024).fit(x, y)
xtd = torch.from_numpy(xt).cuda()
model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize()
res = {}
for m in (1024, 512, 256, 128):
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.predict(xtd[:m].contiguous(), cov="diag", as_numpy=False); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    res[m] = round(sorted(ts)[1], 2)
print(json.dumps({"N": n, "predict_ms_by_test_rows": res, "note": "alpha already solved (no CG inside); level-1 variance"}))
