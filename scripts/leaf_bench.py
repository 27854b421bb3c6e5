"""Back-to-back Cholesky leaves (128 x 128, one workgroup each) on a warm chip: microseconds per launch, by HIP events.
scripts/leaf_bench.py [3]  (3: round 4's column phases, A/B)"""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load(knobs=True)
torch.manual_seed(0)
n, nb = 128, 512
base = torch.randn((nb, n, n), device=G.dev())
spd = base @ base.transpose(1, 2) / n + torch.eye(n, device=G.dev()) * 2.0
dinv = torch.empty((nb, 128, 128), device=G.dev()); cl = torch.zeros(1, dtype=torch.int32, device=G.dev())
if len(sys.argv) > 1: _lib.check(lib.nngp_debug_set(3, int(sys.argv[1])))
a = spd.clone()
out = []
for rep in range(6):
    a.copy_(spd)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(nb):
        _lib.check(lib.nngp_potrf_f32(_lib.ptr(a[b]), n, n, _lib.ptr(dinv[b]), _lib.ptr(cl), _lib.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3 / nb)
l = torch.tril(a).double()
res = float((l @ l.transpose(1, 2) - spd.double()).abs().max())
print("form", sys.argv[1:] , "us per leaf launch:", " ".join("%.1f" % t for t in out), "| residual %.2e clamped %d" % (res, int(cl.item())))
