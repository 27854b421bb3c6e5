import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, gpu_util as G
torch.manual_seed(0)
for k in (96, 160, 928, 992, 1024):
    m, n = 512, 768
    a = torch.randn((m, k), device=G.dev()); b = torch.randn((n, k), device=G.dev()); c0 = torch.randn((m, n), device=G.dev())
    ref = c0.double() - a.double() @ b.double().T
    c = c0.clone(); G.gemm_nt(c, a, b, -1.0, 1.0)
    e32 = float((c.double() - ref).abs().max())
    c = c0.clone(); G.gemm_nt_h3(c, a, b, -1.0, 1.0, 2.0 ** 10)
    eh3 = float((c.double() - ref).abs().max())
    # strided operands (ld > k), like the factorisation's panels
    big_a = torch.randn((m, 1024), device=G.dev()); big_b = torch.randn((n, 1024), device=G.dev())
    av, bv = big_a[:, :k], big_b[:, :k]
    ref2 = c0.double() - av.double() @ bv.double().T
    c = c0.clone(); G.gemm_nt(c, av, bv, -1.0, 1.0)
    e32s = float((c.double() - ref2).abs().max())
    print(k, "f32", e32, "h3", eh3, "f32 strided", e32s)
