"""NTK covariance: the sweep estimate (nngp_model_sweep_estimate) against the error two fixed sweeps really leave, over
the NTK cases of the random parity sweep (tests/test_gpu_parity.py draws the same cases) and two bench-sized fits.
Reference: level 6 (six sweeps, or CG to convergence when the library decides so).  One JSON line per case."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel


def sweep_case(seed):  # the generator of tests/test_gpu_parity.py::_sweep_case
    r = np.random.default_rng(1000 + seed)
    get = "ntk" if seed % 3 == 2 else "nngp"
    n = int(r.integers(40, 1600 if get == "ntk" else 5200))
    return dict(seed=seed, get=get, n=n, m=int(r.integers(1, 400)), d=int(r.choice([2, 3, 7, 20, 64, 128, 200, 256])),
                n_relu=int(r.integers(1, 5)), w=float(r.uniform(0.6, 1.8)), b=float(r.choice([0.0, 0.05, 0.3])),
                diag_reg=float(r.choice([1e-4, 1e-3, 1e-2])), absolute=bool(r.integers(0, 4) == 0), join=bool(r.integers(0, 3) == 0))


def arch(n_relu, w, b):
    return [w] * (n_relu + 1), [b] * (n_relu + 1)


def study(c, x, y, xt, w_std, b_std):
    model = GPModel(c["n"], c["d"], w_std, b_std, get="ntk", diag_reg=c["diag_reg"],
                    diag_reg_absolute_scale=c.get("absolute", False), m_cap=len(xt), knobs=True).fit(x, y)
    model.debug_set(6, 0); model.set_refine(6)
    _, ref = model.predict(xt, cov="diag")
    ref_iters = model.cov_iters()
    model.set_refine(1)
    _, var = model.predict(xt, cov="diag")
    (est_row, est), cov_iters = model.sweep_estimate(), model.cov_iters()
    model.debug_set(6, 1)
    _, fixed = model.predict(xt, cov="diag")
    model.debug_set(6, 0)
    info = model.info()
    rel = lambda v: float(np.max(np.abs(np.asarray(v) - ref) / np.abs(ref)))
    row = dict(c, cg_iters=info["refine_iters"], est=est, est_row=est_row, cov_iters=cov_iters, ref_cov_iters=ref_iters,
               err_default=rel(var), err_two_sweeps=rel(fixed))
    model.close()
    print(json.dumps(row), flush=True)


for seed in range(int(os.environ.get("CASES", "150"))):
    c = sweep_case(seed)
    if c["get"] != "ntk":
        continue
    if c["absolute"]:
        c["diag_reg"] *= 1e5
    jb = c["join"] and c["d"] >= 8
    x, y = synth.synthetic_queries(c["n"], c["d"], seed=seed, join_block=jb)
    xt, _ = synth.synthetic_queries(c["m"], c["d"], seed=seed + 100, join_block=jb)
    w_std, b_std = arch(c["n_relu"], c["w"], c["b"])
    study(c, x, y, xt, w_std, b_std)
for n, d, reg in ((16384, 256, 1e-3), (8192, 64, 1e-3), (8192, 256, 1e-4), (4096, 20, 1e-3)):
    c = dict(seed=-1, get="ntk", n=n, m=1024, d=d, n_relu=1, w=1.0, b=0.0, diag_reg=reg, absolute=False, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0, join_block=True)
    xt, _ = synth.synthetic_queries(1024, d, seed=1, join_block=True)
    study(c, x, y, xt, [1.0, 1.0], [0.0, 0.0])
