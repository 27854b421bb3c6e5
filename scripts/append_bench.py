"""Active-learning update cost: GPModel.append (extend the factor) + solve against a full refit, same final training set."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

out = {}
for n0, b, d, n_relu in [(7800, 1000, 20, 1), (9800, 1000, 20, 1), (31744, 1024, 128, 3)]:
    n1 = n0 + b
    x, y = synth.synthetic_queries(n1, d, seed=0)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    w, bb = [1.0] * (n_relu + 1), [0.0] * (n_relu + 1)
    model = GPModel(n1, d, w, bb, diag_reg=1e-3)
    # (info() forces the deferred CG solve, so that every timing covers alpha)
    full = timed(lambda: (model.fit(xd, yd), model.info()))
    a_full = model.alpha().clone(); it_full = model.info()["refine_iters"]
    def inc():
        model.fit(xd[:n0], yd[:n0]); model.info()
    base = timed(inc)
    def inc2():
        model.fit(xd[:n0], yd[:n0]); model.info(); model.append(xd[n0:], yd[n0:]); model.info()
    both = timed(inc2)
    a_inc = model.alpha(); info = model.info()
    out["N%d+%d" % (n0, b)] = {"full_refit_ms": round(full, 2), "append_plus_solve_ms": round(both - base, 2),
                               "cg_iters_full": it_full, "cg_iters_append": info["refine_iters"],
                               "alpha_rel_diff": float(torch.linalg.vector_norm(a_inc - a_full) / torch.linalg.vector_norm(a_full))}
    model.close(); del model
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
