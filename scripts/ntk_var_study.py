"""NTK posterior variance at cfg5 (N=16384, d=256, join block): error and time of 1 / 2 / 3 fixed correction sweeps
(adaptive continuation off: timing-knob key 6 = 1) and of the adaptive default, against 4 sweeps."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d, m = int(os.environ.get("N", 16384)), 256, 1024
x, y = synth.synthetic_queries(n, d, seed=0, join_block=True)
xt, _ = synth.synthetic_queries(m, d, seed=1, join_block=True)
model = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], get="ntk", diag_reg=1e-3, m_cap=m, knobs=True)
model.fit(x, y)
def run(level, fixed):
    model.debug_set(6, 1 if fixed else 0)
    model.set_refine(level)
    model.predict(xt, cov="diag"); torch.cuda.synchronize()
    t0 = time.perf_counter(); mean, var = model.predict(xt, cov="diag"); torch.cuda.synchronize()
    return np.asarray(var, dtype=np.float64), (time.perf_counter() - t0) * 1e3
ref, _ = run(4, True)
out = {"info": {k: (float(v) if isinstance(v, float) else int(v)) for k, v in model.info().items() if isinstance(v, (int, float))}}
for name, level, fixed in (("1_sweep_fixed", 1, True), ("2_sweeps_fixed", 2, True), ("3_sweeps_fixed", 3, True), ("default_adaptive", 1, False)):
    v, ms = run(level, fixed)
    out[name] = {"max_rel_err": float(np.max(np.abs(v - ref) / np.abs(ref))), "predict_ms": round(ms, 2)}
print(json.dumps(out, indent=1))
