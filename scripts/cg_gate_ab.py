"""cfg3-like predict with the deferred alpha CG gated on the start of the int8 plane products (default) against the ungated start
beside the blocked solves (timing-knob key 2 = 8), gated on their end (key 2 = 9), and the predict alone (alpha already there)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
cfgs = {"cfg3": (32768, 128, 1024, 3), "cfg2": (8192, 64, 1024, 1), "forest": (10800, 20, 3600, 1), "cfg4": (65536, 128, 1024, 3)}
for name in sys.argv[1:] or ["cfg3"]:
    n, d, m, relu = cfgs[name]
    x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
    xd, yd, xtd = (torch.from_numpy(a).cuda() for a in (x, y, xt))
    model = GPModel(n, d, [1.0] * (relu + 1), [0.0] * (relu + 1), diag_reg=1e-3, m_cap=m, knobs=True)
    def timed(f):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
    res = {"config": name, "gated": [], "ungated": [], "gated_at_end": [], "alpha_first": []}
    for rep in range(5):
        for key, tag in ((0, "gated"), (8, "ungated"), (9, "gated_at_end")):
            model.debug_set(2, key)
            model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve()
            res[tag].append(round(timed(lambda: model.predict(xtd, cov="diag", as_numpy=False)), 2))
        model.debug_set(2, 0)
        model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve(); model.alpha()
        res["alpha_first"].append(round(timed(lambda: model.predict(xtd, cov="diag", as_numpy=False)), 2))
    print(json.dumps(res), flush=True)
    model.close()
