"""Factor of a fixed float32 SPD matrix (N = 3072) through nngp_potrf_f32: SHA-256 of L and the inverted diagonal blocks, and the time per call.
Run once plain and once with NNGP_POTRF_NO_SIDE=1: the hashes must agree (the look-ahead inside the recursion moves row blocks of the same
kernels to a side stream; it changes no arithmetic)."""
import hashlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
g = torch.Generator(device="cpu").manual_seed(5)
base = torch.randn((n, n), generator=g)
spd = (base @ base.T / n + torch.eye(n) * 0.5).to(G.dev())
dinv = torch.empty((n // 128, 128, 128), device=G.dev()); cl = torch.zeros(1, dtype=torch.int32, device=G.dev())
ts = []
for rep in range(6):
    a = spd.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.nngp_potrf_f32(_lib.ptr(a), n, n, _lib.ptr(dinv), _lib.ptr(cl), _lib.stream_ptr()))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
L = torch.tril(a).cpu().numpy()
h = hashlib.sha256(L.tobytes() + dinv.cpu().numpy().tobytes()).hexdigest()[:16]
res = float((torch.tril(a).double() @ torch.tril(a).double().T - spd.double()).abs().max())
print("side" if not os.environ.get("NNGP_POTRF_NO_SIDE") else "plain", "sha", h, "residual %.2e" % res, "ms", " ".join("%.3f" % t for t in ts), "clamped", int(cl.item()))
