#!/usr/bin/env python3
"""Time of the alpha CG alone on an idle GPU (solve + info on a fitted model)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d, n_relu = [int(v) for v in sys.argv[1:4]]
x, y = synth.synthetic_queries(n, d, seed=0)
model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=128).fit(x, y)
model.info()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.solve(); info = model.info()
    torch.cuda.synchronize(); print("CG alone: %.2f ms, %d iterations" % ((time.perf_counter() - t0) * 1e3, info["refine_iters"]), flush=True)
