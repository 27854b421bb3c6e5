"""cfg3: how long is the covariance + mean part of predict with alpha already there (no CG beside it), against the
predict that overlaps the deferred CG -- the difference is what the CG still costs on the critical path."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d, m = int(os.environ.get("N", 32768)), 128, 1024
x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
xd, yd, xtd = (torch.from_numpy(a).cuda() for a in (x, y, xt))
knobs = bool(os.environ.get("NNGP_DEBUG"))
if knobs:
    from nngp_src_amd import _lib
    for kv in os.environ["NNGP_DEBUG"].split(","):
        k, v = kv.split("=")
        _lib.load(knobs=True).nngp_debug_set(int(k), int(v))
model = GPModel(n, d, [1.0] * 4, [0.0] * 4, diag_reg=1e-3, m_cap=m, knobs=knobs)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
res = {"with_cg": [], "alpha_first": [], "cg_alone": []}
for rep in range(4):
    model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve()
    res["with_cg"].append(round(timed(lambda: model.predict(xtd, cov="diag", as_numpy=False)), 2))
    model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve()
    res["cg_alone"].append(round(timed(lambda: model.alpha()), 2))
    res["alpha_first"].append(round(timed(lambda: model.predict(xtd, cov="diag", as_numpy=False)), 2))
print(json.dumps(res))
