"""Phase stamps of the Cholesky leaf (timing-knob build, debug key 7 = 8): one 128x128 factorisation, printed to stderr.
scripts/leaf_stamps.py [3]  (3: round 4's column phases, A/B)"""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load(knobs=True)
torch.manual_seed(0)
if len(sys.argv) > 1: _lib.check(lib.nngp_debug_set(3, int(sys.argv[1])))  # 3: round 4's column phases (A/B)
n = 128
base = torch.randn((n, n), device=G.dev())
spd = base @ base.T / n + torch.eye(n, device=G.dev()) * 2.0
dinv = torch.empty((1, 128, 128), device=G.dev()); cl = torch.zeros(1, dtype=torch.int32, device=G.dev())
for rep in range(3):
    a = spd.clone()
    if rep == 2: _lib.check(lib.nngp_debug_set(7, 8))
    _lib.check(lib.nngp_potrf_f32(_lib.ptr(a), n, n, _lib.ptr(dinv), _lib.ptr(cl), _lib.stream_ptr()))
    torch.cuda.synchronize()
_lib.check(lib.nngp_debug_set(7, 0))
l = torch.tril(a).double(); print("residual", float((l @ l.T - spd.double()).abs().max()), "inverse", float((torch.tril(dinv[0]).double() @ l - torch.eye(n, device=G.dev(), dtype=torch.float64)).abs().max()))
