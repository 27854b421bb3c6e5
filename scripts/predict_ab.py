import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
#!/usr/bin/env python3
"""A/B of one predict configuration under a debug key (NNGP_AB="key=value"): time per call, back to back."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel
n, d, n_relu, m = [int(v) for v in sys.argv[1:5]]
x, y = synth.synthetic_queries(n, d, seed=0)
model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=1024).fit(x, y)
xt, _ = synth.synthetic_queries(m, d, seed=1)
xtd = torch.from_numpy(xt).cuda()
for tag, kv in [("default", None)] + [(a, a) for a in sys.argv[5:]]:
    for k in range(8):
        _lib.load().nngp_debug_set(k, 0)
    if kv:
        k, v = kv.split("=")
        _lib.load().nngp_debug_set(int(k), int(v))
    for rep in range(2):
        model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.predict(xtd, cov="diag", as_numpy=False)
        torch.cuda.synchronize()
        print(tag, "ms/call", round((time.perf_counter() - t0) / 10 * 1e3, 3), flush=True)
