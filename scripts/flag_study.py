#!/usr/bin/env python3
"""Where does the row flag of the adaptive covariance (k_rows_prepare) fire?  Sweeps its threshold (debug key 6) at a
bench-sized problem and reports whether predict went on by CG (cov_iters) and what that changed."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel

n, d, n_relu, m = [int(v) for v in sys.argv[1:5]]
x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=m).fit(x, y)
lib = _lib.load()
out = {"cg_iters": model.info()["refine_iters"]}
lib.nngp_debug_set(6, 1)
model.set_refine(4); ref = model.predict(xt, cov="diag")[1]
model.set_refine(2)
for e in (1, 5, 6, 7, 8, 9, 10, 12):
    lib.nngp_debug_set(6, e)
    for cov in ("diag", "full"):
        xs = xt if cov == "diag" else xt[:256]
        model.predict(xs, cov=cov)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, v = model.predict(xs, cov=cov)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        v = v if cov == "diag" else np.diag(v)
        out["thr1e-%d_%s" % (e, cov)] = {"cov_iters": model.cov_iters(), "ms": round(dt, 2),
                                         "max_rel_vs_level4": float(np.max(np.abs(v - ref[:len(v)]) / ref[:len(v)]))}
print(json.dumps(out, indent=1))
