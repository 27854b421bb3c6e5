"""The guard's estimate (nngp_model_residual_floor) over sizes and conditionings: (N, d, layers, diag_reg) -> estimate, whether the fit was
taken off the int8 path, CG iterations, and the measured distance of the level-1 variances from level 3 (float64 products)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
cases = [(4096, 64, 1, 1e-3), (8192, 64, 1, 1e-3), (16384, 128, 3, 1e-3), (32768, 128, 3, 1e-3), (65536, 128, 3, 1e-3),
         (10800, 20, 1, 1e-3), (8100, 3, 3, 1e-4), (12000, 7, 2, 1e-4), (6000, 2, 3, 1e-2), (20000, 20, 1, 1e-4)]
for (n, d, relu, reg) in cases:
    x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(512, d, seed=1)
    model = GPModel(n, d, [1.0] * (relu + 1), [0.0] * (relu + 1), diag_reg=reg, m_cap=512, knobs=True).fit(x, y)
    _, v1 = model.predict(xt, cov="diag")
    ratio, distrusted = model.residual_floor()
    it = model.info()["refine_iters"]; cov_it = model.cov_iters()
    model.debug_set(5, 50); model.set_refine(3); _, v3 = model.predict(xt, cov="diag"); model.debug_set(5, 0)
    print(json.dumps({"N": n, "d": d, "n_relu": relu, "diag_reg": reg, "floor_estimate": ratio, "distrusted": distrusted, "cg_iters": it,
                      "cov_iters": cov_it, "level1_vs_level3": float(np.max(np.abs(v1 - v3) / np.abs(v3)))}), flush=True)
    model.close()
