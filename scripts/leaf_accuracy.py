"""Quality of the float32 factor on the ill-conditioned NTK fits of the random sweep, round 4's leaf (timing-knob key 3 = 3) against round 5's:
backward error |L L^T - A| / |A|, forward error against a float64 Cholesky of the same float32 matrix, and the preconditioned spectrum
|I - L^-1 A L^-T|_2 (what the CG sees).  scripts/leaf_accuracy.py [seeds...]"""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import sys
import numpy as np, torch, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gpu_util as G
import nngp_oracle as o
from nngp_src_amd import _lib, synth
from test_gpu_parity import _sweep_case
lib = _lib.load(knobs=True)
seeds = [int(s) for s in sys.argv[1:]] or [29, 83, 146]
for seed in seeds:
    c = _sweep_case(seed)
    jb = c["join"] and c["d"] >= 8
    x, y = synth.synthetic_queries(c["n"], c["d"], seed=seed, join_block=jb)
    a = o.make_arch(c["n_relu"], c["w"], c["b"])
    k = G.kernel_build(x, None, a.w_std, a.b_std, get=(c["get"],))[c["get"]]
    n = c["n"]; npad = -(-n // 128) * 128
    reg = c["diag_reg"] * np.trace(k) / n
    A = np.eye(npad, dtype=np.float64) * float(np.mean(np.diag(k)) + reg)
    A[:n, :n] = k + reg * np.eye(n)
    A32 = A.astype(np.float32)
    A64 = A32.astype(np.float64)
    L64 = np.linalg.cholesky(A64)
    print("seed %d n %d cond(A) %.2e" % (seed, n, np.linalg.cond(A64)))
    for form, name in ((3, "round 4 leaf"), (0, "round 5 leaf")):
        _lib.check(lib.nngp_debug_set(3, form))
        ad = torch.from_numpy(A32.copy()).to(G.dev())
        dinv = torch.empty((npad // 128, 128, 128), device=G.dev()); cl = torch.zeros(1, dtype=torch.int32, device=G.dev())
        _lib.check(lib.nngp_potrf_f32(_lib.ptr(ad), npad, npad, _lib.ptr(dinv), _lib.ptr(cl), _lib.stream_ptr()))
        torch.cuda.synchronize()
        L = np.tril(ad.cpu().numpy().astype(np.float64))
        bwd = np.linalg.norm(L @ L.T - A64) / np.linalg.norm(A64)
        fwd = np.linalg.norm(L - L64) / np.linalg.norm(L64)
        Y = sla.solve_triangular(L, A64, lower=True)
        Mm = sla.solve_triangular(L, Y.T, lower=True)
        spec = np.linalg.norm(np.eye(npad) - Mm, 2)
        # the exact float64 kernel (what the CG iterates on) through the same preconditioner
        Y2 = sla.solve_triangular(L, A, lower=True); M2 = sla.solve_triangular(L, Y2.T, lower=True)
        ev = np.linalg.eigvalsh(0.5 * (M2 + M2.T))
        print("  %-13s clamped %d  backward %.3e  forward %.3e  |I - L^-1 A32 L^-T|_2 %.3e  eig(L^-1 A64 L^-T) in [%.4f, %.4f]" % (name, int(cl.item()), bwd, fwd, spec, ev.min(), ev.max()))
_lib.check(lib.nngp_debug_set(3, 0))
