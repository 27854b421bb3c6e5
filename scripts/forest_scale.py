"""Stage times at the reference's default run size (forest: N=10800 train, M=3600 test, d=20, 3-layer ReLU NNGP)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, m, d = 10800, 3600, 20
x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
xd, yd, xtd = (torch.from_numpy(a).cuda() for a in (x, y, xt))
model = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, m_cap=m)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); return e
res = {}
for rep in range(4):
    e = [ev()]
    model.set_train(xd, yd); e.append(ev())
    model.build_rows(0, n); e.append(ev())
    model.factor(); e.append(ev())
    model.solve(); e.append(ev())
    out = model.predict(xtd, cov="diag", as_numpy=False); e.append(ev())
    torch.cuda.synchronize()
    res = {k: round(e[i].elapsed_time(e[i + 1]), 3) for i, k in enumerate(["set_train", "build", "cholesky", "solve_request", "posterior+solve"])}
    res["total"] = round(e[0].elapsed_time(e[-1]), 3)
for level in (0, 1, 2):
    model.set_refine(level)
    model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize()
    res["predict_level%d_ms" % level] = round((time.perf_counter() - t0) * 1e3, 3)
res["cg_iters"] = model.info()["refine_iters"]
print(json.dumps(res))
