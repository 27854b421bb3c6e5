"""Times the sliced int8 product (csrc/gemm_i8s.hip) at the posterior's residual shape through the C ABI test entry; run under
rocprofv3 --kernel-trace --stats for per-kernel times.  usage: i8s_lab.py [m n k [sa sb cut [reps]]]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gpu_util as G
a = [int(v) for v in sys.argv[1:]]
m, n, k = (a + [1024, 32768, 32768])[:3] if len(a) >= 3 else (1024, 32768, 32768)
sa, sb, cut = a[3:6] if len(a) >= 6 else (5, 5, 4)
reps = a[6] if len(a) >= 7 else 3
torch.manual_seed(0)
A = torch.randn((m, k), device=G.dev(), dtype=torch.float64) * torch.exp2(torch.randint(-12, 1, (m, k), device=G.dev()).double())
B = torch.randn((n, k), device=G.dev(), dtype=torch.float64)
C = torch.empty((m, n), device=G.dev(), dtype=torch.float64)
C64 = torch.empty_like(C)
out = {"m": m, "n": n, "k": k, "slices": [sa, sb, cut]}
for name, fn in (("i8s", lambda: G.gemm_nt_i8s(C, None, A, B, 1.0, 0.0, sa, sb, cut)), ("f64", lambda: G.gemm_nt_f64(C64, None, A, B, 1.0, 0.0))):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    out[name + "_ms_wall_min"] = min(ts)
unit = A.abs().amax(1)[:, None] * B.abs().amax(1)[None, :]
out["max_err_vs_f64_rel_unit"] = float(((C - C64).abs() / unit).max())
print(json.dumps(out))
