"""Backward error of the float32 Cholesky factor, || L L^T - (K + reg I) ||_F / || K + reg I ||_F, for the split-float16
trailing updates (default) and the float32-MFMA ones (debug key 2 = 2).  Float64 reference product on the GPU."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel

out = {}
n, d, n_relu = int(os.environ.get("FR_N", "8192")), 128, 3
x, y = synth.synthetic_queries(n, d, seed=0)
for key2 in (0, 2):
    _lib.load().nngp_debug_set(2, key2)
    model = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3).fit(x, y)
    info = model.info()
    k64, ld = model.kernel_buffer()
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double()
    A = k64[:n, :n].clone()
    A.diagonal().add_(info["reg"])
    R = L @ L.T - A
    out["h3" if key2 == 0 else "f32"] = {"backward_rel_fro": float(R.norm() / A.norm()), "max_abs_over_maxdiag": float(R.abs().max() / A.diagonal().max()),
                                         "cg_iters": info["refine_iters"], "clamped": info["clamped_pivots"]}
    model.close(); del model, L, A, R
    torch.cuda.empty_cache()
_lib.load().nngp_debug_set(2, 0)
print(json.dumps(out, indent=1))
