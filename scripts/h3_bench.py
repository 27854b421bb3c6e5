"""Accuracy and timing of the split-float16 GEMM against the float32-MFMA GEMM (run under rocprofv3 for kernel times)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as G
from nngp_src_amd import _lib

dev = G.dev()
torch.manual_seed(0)
out = {}
for (m, n, k, lower) in [(1024, 768, 512, False), (4096, 4096, 1024, False), (16384, 16384, 1024, True), (16384, 16384, 1024, False),
                         (32768 - 2048, 32768 - 2048, 1024, True)]:
    a = torch.randn((m, k), device=dev) * torch.exp2(torch.randint(-6, 1, (m, 1), device=dev).float())
    b = a if lower else torch.randn((n, k), device=dev)
    c0 = torch.randn((m, n), device=dev)
    ref = None
    if m <= 4096:
        ref = c0.double() - a.double() @ b.double().T
    res = {}
    for name in ("f32", "h3"):
        c = c0.clone()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for rep in range(3):
            if rep == 2: e0.record()
            if name == "f32":
                G.gemm_nt(c, a, b, -1.0, 1.0, lower)
            else:
                G.gemm_nt_h3(c, a, b, -1.0, 1.0, 2.0 ** 10, lower)
        e1.record(); torch.cuda.synchronize()
        res[name + "_ms_with_overheads"] = e0.elapsed_time(e1)
        if ref is not None:
            c = c0.clone()
            (G.gemm_nt if name == "f32" else (lambda *x: G.gemm_nt_h3(*x[:5], 2.0 ** 10, x[5])))(c, a, b, -1.0, 1.0, lower)
            res[name + "_max_err"] = float((c.double() - ref).abs().max())
            res[name + "_rel_fro"] = float((c.double() - ref).norm() / ref.norm())
        else:
            res[name + "_sum"] = float(torch.tril(c).double().sum()) if lower else float(c.double().sum())
    out["%dx%dx%d%s" % (m, n, k, "_lower" if lower else "")] = res
    del a, b, c0, c
print(json.dumps(out, indent=1))
