#!/usr/bin/env python3
"""Serving-path latency: predict(x_test) wall time per query batch on a fitted model, through the
solve path (float32 blocked solves + float64 sweeps) and in serving mode (explicit float64 inverse, nngp_model_prepare_serving)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel


def timed(model, xtd, reps=5):
    model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.predict(xtd, cov="diag", as_numpy=False)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)


out = {}
for n, d, n_relu in [(10800, 20, 1), (32768, 128, 3)]:
    x, y = synth.synthetic_queries(n, d, seed=0)
    model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=1024).fit(x, y)
    res, ref = {}, {}
    for m in (1, 8, 16, 128, 1024):
        xt, _ = synth.synthetic_queries(m, d, seed=1)
        xtd = torch.from_numpy(xt).cuda()
        for level in (0, 2):
            model.set_refine(level)
            res["M%d_level%d_solve_ms" % (m, level)] = timed(model, xtd)
        model.set_refine(3)
        ref[m] = model.predict(xtd, cov="diag")[1]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.prepare_serving()
    torch.cuda.synchronize(); res["prepare_serving_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    model.set_refine(2)
    for m in (1, 8, 16, 128, 1024):
        xt, _ = synth.synthetic_queries(m, d, seed=1)
        xtd = torch.from_numpy(xt).cuda()
        res["M%d_serving_ms" % m] = timed(model, xtd)
        v = model.predict(xtd, cov="diag")[1]
        res["M%d_serving_var_max_rel_vs_level3" % m] = float(np.max(np.abs(v - ref[m]) / ref[m]))
    out["N%d" % n] = res
    model.close(); del model; torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
