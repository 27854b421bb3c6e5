import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d, n_relu, m = 32768, 128, 3, 1024
x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
xd, yd, xtd = (torch.from_numpy(a).cuda() for a in (x, y, xt))
model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, m_cap=m)
def step(tol):
    model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve(0, tol)
    return model.predict(xtd, cov="diag", as_numpy=False)
ref = step(1e-13)[0].cpu().numpy(); info = model.info(); print("ref iters", info["refine_iters"], info["rel_residual"])
for tol in (1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
    step(tol); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = step(tol)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    mean = out[0].cpu().numpy(); info = model.info()
    print("tol %.0e: %.1f ms/step, iters %d, relres %.1e, mean rel l2 %.1e, max elem %.1e" % (tol, dt, info["refine_iters"], info["rel_residual"],
          np.linalg.norm(mean - ref) / np.linalg.norm(ref), np.max(np.abs(mean - ref) / np.maximum(1, np.abs(ref)))))
