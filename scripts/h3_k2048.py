"""Would a two-panel (K = 2048) trailing update beat two K = 1024 updates?  Same 30720-row lower update, kernel times from
a rocprofv3 kernel trace (run under scripts/gpu_h3_k2048.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as G
torch.manual_seed(0)
m = int(os.environ.get("M", 30720))
a1 = torch.randn((m, 1024), device=G.dev()); a2 = torch.randn((m, 2048), device=G.dev())
c = torch.zeros((m, m), device=G.dev())
for rep in range(6):
    G.gemm_nt_h3(c, a1, a1, -1.0, 1.0, 2.0 ** 8, True)
    G.gemm_nt_h3(c, a2, a2, -1.0, 1.0, 2.0 ** 8, True)
