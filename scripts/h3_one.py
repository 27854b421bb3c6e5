"""One split-float16 trailing-update shape (30720 rows, K = 1024, lower), a few launches: target of rocprofv3 --pmc runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as G
torch.manual_seed(0)
for (m, k) in ((30720, 1024), (24576, 4096)):  # a K = 1024 update, and the deep-K shape of the grouped Cholesky's far updates
    a = torch.randn((m, k), device=G.dev())
    c = torch.zeros((m, m), device=G.dev())
    for rep in range(3):
        G.gemm_nt_h3(c, a, a, -1.0, 1.0, 2.0 ** 10, True)
    del a, c
