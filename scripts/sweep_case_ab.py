"""One case of the large-N parity sweep under two Cholesky schedules (debug key 2 = 11: one block column per group; 0: grouped):
level-1 variance, level-2 diag of the full covariance, float64 C oracle."""
import os, sys, json
os.environ.setdefault("NNGP_SWEEP_NMIN", "5000"); os.environ.setdefault("NNGP_SWEEP_NMAX", "14000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import test_gpu_parity as T
from nngp_src_amd import _lib, synth
from nngp_src_amd.model import GPModel
import nngp_oracle as o
import c_oracle
seed = int(sys.argv[1])
c = T._sweep_case(seed)
if c["absolute"]:
    c["diag_reg"] *= 1e5
x, y = synth.synthetic_queries(c["n"], c["d"], seed=seed, join_block=c["join"] and c["d"] >= 8)
xt, _ = synth.synthetic_queries(c["m"], c["d"], seed=seed + 100, join_block=c["join"] and c["d"] >= 8)
a = o.make_arch(c["n_relu"], c["w"], c["b"])
ref = c_oracle.fit(x, y, a.w_std, a.b_std, diag_reg=c["diag_reg"], absolute=c["absolute"])
mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
lib = _lib.load(knobs=True)
print(json.dumps(c))
for key2 in (11, 0):
    lib.nngp_debug_set(2, key2)
    m = GPModel(c["n"], c["d"], a.w_std, a.b_std, get="nngp", diag_reg=c["diag_reg"], diag_reg_absolute_scale=c["absolute"], knobs=True).fit(x, y)
    mean, var = m.predict(xt, cov="diag")
    ci = m.cov_iters()
    info = m.info()
    _, cov = m.predict(xt[:64], cov="full")
    d12 = np.abs(np.diag(cov) - var[:64]).max() / np.abs(var[:64]).max()
    v1 = np.max(np.abs(var - var_ref.ravel()) / np.maximum(np.abs(var_ref.ravel()), 1e-9 * np.abs(var_ref).max()))
    v2 = np.max(np.abs(np.diag(cov) - var_ref.ravel()[:64]) / np.maximum(np.abs(var_ref.ravel()[:64]), 1e-9 * np.abs(var_ref).max()))
    print("key2=%d cg_iters %d cov_iters %d  level1-vs-level2 %.2e  level1-vs-oracle %.2e  level2-vs-oracle %.2e" % (key2, info["refine_iters"], ci, d12, v1, v2))
    m.close()
lib.nngp_debug_set(2, 0)
