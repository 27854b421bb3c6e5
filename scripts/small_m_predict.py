"""Predict latency for a SMALL block of test rows against a large fit (the multi-GPU layout shards M = 1024 test rows over the
ranks: 128 per rank at 8 GPUs): split-float16 solve path (default when mp * np >= 7e6) against the float32 one (timing-knob key
7 = 2)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d = int(os.environ.get("N", 65536)), 128
x, y = synth.synthetic_queries(n, d, seed=0)
model = GPModel(n, d, [1.0] * 4, [0.0] * 4, diag_reg=1e-3, m_cap=1024, knobs=True)
model.fit(x, y); model.info()
out = {}
for m in (128, 256, 512, 1024):
    xt, _ = synth.synthetic_queries(m, d, seed=1)
    for name, key in (("h3", 0), ("f32", 2)):
        model.debug_set(7, key)
        model.predict(xt, cov="diag"); torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); mean, var = model.predict(xt, cov="diag"); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out["M%d_%s" % (m, name)] = round(min(ts), 2)
        if name == "h3": ref = np.asarray(var)
        else: out["M%d_var_diff" % m] = float(np.max(np.abs(np.asarray(var) - ref) / np.abs(ref)))
    print(m, out, flush=True)
model.debug_set(7, 0)
print(json.dumps(out))
