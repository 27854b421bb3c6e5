#!/usr/bin/env python3
"""Accuracy / cost of the posterior-variance precision levels at sizes too big for a CPU oracle (run on the GPU box).
Level 4 (three correction sweeps + second-order formula) is the reference; reports the max relative deviation of var at
levels 0..3 and timings."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel

out = {}
SIZES = [(4096, 64, 1, 512), (16384, 128, 3, 512), (32768, 128, 3, 1024)]
if os.environ.get("VAR_STUDY_N"):
    SIZES = [t for t in SIZES if t[0] == int(os.environ["VAR_STUDY_N"])]
if os.environ.get("NNGP_DEBUG"):  # e.g. "2=2" float32-MFMA Cholesky, "7=2" float32 blocked solves
    from nngp_src_amd import _lib
    for kv in os.environ["NNGP_DEBUG"].split(","):
        k, v = kv.split("=")
        _lib.load(knobs=True).nngp_debug_set(int(k), int(v))
for n, d, n_relu, m in SIZES:
    x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
    model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=m, knobs=True).fit(x, y)
    res = {}
    for level in (4, 0, 1, "1old", 2, 3):  # "1old": full float64 residual + preconditioned remainder (knob 5 = 1)
        model.set_refine(1 if level == "1old" else level)
        model.debug_set(5, 1 if level == "1old" else 0)
        model.predict(xt, cov="diag")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mean, var = model.predict(xt, cov="diag")
        torch.cuda.synchronize(); res[level] = (var, (time.perf_counter() - t0) * 1e3, mean)
    model.debug_set(5, 0)
    ref = res[4][0]
    out["N%d" % n] = {"cg_iters": model.info()["refine_iters"], "var_min": float(ref.min()), "var_median": float(np.median(ref)),
                      **{"level%s" % l: {"max_rel": float(np.max(np.abs(res[l][0] - ref) / ref)), "ms": round(res[l][1], 2),
                                         "mean_rel_l2_vs_level4": float(np.linalg.norm(res[l][2] - res[4][2]) / np.linalg.norm(res[4][2]))}
                         for l in (0, 1, "1old", 2, 3)}}
    model.close(); del model
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
