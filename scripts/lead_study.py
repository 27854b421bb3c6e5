"""How many leading columns of block column 0 have to stay on the float32 MFMA (accumulator truncation of the float16 pipe on
same-sign sums)?  Timing-knob key 3 = 30 + c: lead = 32 c columns (20 + c: 128 c).  Per variant: Cholesky ms, CG iterations,
worst relative error of the default (level 1) variance against level 3 computed with the default factor."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d, m = int(os.environ.get("N", 32768)), 128, 1024
x, y = synth.synthetic_queries(n, d, seed=0)
xt, _ = synth.synthetic_queries(m, d, seed=1)
model = GPModel(n, d, [1.0] * 4, [0.0] * 4, diag_reg=1e-3, m_cap=m, knobs=True)
model.fit(x, y); model.set_refine(3); _, ref = model.predict(xt, cov="diag"); ref = np.asarray(ref)
out = {}
for name, key in (("lead256", 0), ("lead128", 21), ("lead96", 33), ("lead64", 32), ("lead32", 31), ("lead0", 20)):
    model.debug_set(3, key)
    model.set_refine(1)
    ts = []
    for rep in range(3):
        model.set_train(x, y); model.build_rows(0, n); torch.cuda.synchronize()
        t0 = time.perf_counter(); model.factor(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    model.solve()
    _, var = model.predict(xt, cov="diag")
    info = model.info()
    out[name] = {"cholesky_ms": round(min(ts), 2), "cg_iters": int(info["refine_iters"]), "rel_residual": float(info["rel_residual"]),
                 "var_level1_vs_level3": float(np.max(np.abs(np.asarray(var) - ref) / np.abs(ref)))}
    print(name, out[name], flush=True)
model.debug_set(3, 0)
print(json.dumps(out, indent=1))
