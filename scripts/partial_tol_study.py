import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
#!/usr/bin/env python3
"""Early-stopped alpha CG: stopping tolerance (debug key 3 = 40 + e -> 10^-e) against step time and the error of the corrected
mean mu = K_td a_k + Z r_k relative to the fully converged solve."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel
n, d, n_relu, m = [int(v) for v in sys.argv[1:5]]
x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
xd, yd, xtd = (torch.from_numpy(a).cuda() for a in (x, y, xt))
model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=m)
lib = _lib.load()
def step():
    model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve()
    return model.predict(xtd, cov="diag", as_numpy=False)
lib.nngp_debug_set(0, 128)
ref = step()[0].cpu().numpy()
lib.nngp_debug_set(0, 0)
for e in (8, 6, 5, 4, 3, 2, 1):
    lib.nngp_debug_set(3, 40 + e)
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    mean = out[0].cpu().numpy()
    print("stop at 1e-%d: %.1f ms/step, mean rel l2 %.1e, max elem %.1e" % (e, dt, np.linalg.norm(mean - ref) / np.linalg.norm(ref),
          np.max(np.abs(mean - ref) / np.maximum(1, np.abs(ref)))), flush=True)
