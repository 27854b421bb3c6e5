"""Multiplicative truncation bias of the float16 MFMA accumulation vs K, on same-sign and mixed-sign data (float32 MFMA beside it)."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
torch.manual_seed(0)
dev = G.dev()
m = 2048
for kind in ("pos", "mixed"):
    for k in (128, 512, 1024, 4096):
        a = torch.rand((m, k), device=dev) + 0.1
        b = torch.rand((m, k), device=dev) + 0.1
        if kind == "mixed":
            a = a * torch.sign(torch.randn((m, k), device=dev)); b = b * torch.sign(torch.randn((m, k), device=dev))
        ref = a.double() @ b.double().T
        out = {}
        for name in ("f32", "h3"):
            c = torch.zeros((m, m), device=dev)
            if name == "f32": G.gemm_nt(c, a, b, 1.0, 0.0)
            else: G.gemm_nt_h3(c, a, b, 1.0, 0.0, 2.0 ** 13)
            e = c.double() - ref
            # multiplicative bias estimate: regression of e on ref
            out[name] = "%.2e" % float((e * ref).sum() / (ref * ref).sum())
        print(kind, k, out)
