"""Timing of the float64 MFMA GEMM on the posterior's residual shape (M x N x N), torch events, interleaved knob variants."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load(knobs=True)
dev = G.dev()
m, n = int(os.environ.get("F64_M", "1024")), int(os.environ.get("F64_N", "32768"))
a = torch.randn((m, n), dtype=torch.float64, device=dev)
b = torch.randn((n, n), dtype=torch.float64, device=dev)
c = torch.zeros((m, n), dtype=torch.float64, device=dev)
res = {}
for rnd in range(4):
    for v in [int(x) for x in os.environ.get("F64_VARIANTS", "0,6").split(",")]:
        _lib.check(lib.nngp_debug_set(5, v))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.nngp_gemm_nt_f64(_lib.ptr(c), n, _lib.ptr(c), n, _lib.ptr(a), n, _lib.ptr(b), n, m, n, n, -1.0, 0.0, _lib.stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        if rnd > 0:
            res.setdefault(v, []).append(e0.elapsed_time(e1))
_lib.check(lib.nngp_debug_set(5, 0))
for v, t in res.items():
    med = sorted(t)[len(t) // 2]
    print("knob5=%d  ms %s  median %.2f  TF/s %.1f" % (v, ["%.2f" % x for x in t], med, 2.0 * m * n * n / med / 1e9))
