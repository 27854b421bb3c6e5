#!/usr/bin/env python3
"""Serving-path latency: predict(x_test) wall time per query batch on a fitted model."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel

out = {}
for n, d, n_relu in [(10800, 20, 1), (32768, 128, 3)]:
    x, y = synth.synthetic_queries(n, d, seed=0)
    model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=1024).fit(x, y)
    for variant in (0, 1):
        _lib.load().nngp_debug_set(7, variant)
        for m in (1, 16, 128, 1024):
            xt, _ = synth.synthetic_queries(m, d, seed=1)
            xtd = torch.from_numpy(xt).cuda()
            for level in (0, 2):
                model.set_refine(level)
                model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    model.predict(xtd, cov="diag", as_numpy=False)
                torch.cuda.synchronize()
                out["N%d_M%d_level%d_%s" % (n, m, level, "rec128" if variant else "blk1024")] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    _lib.load().nngp_debug_set(7, 0)
    model.close(); del model; torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
