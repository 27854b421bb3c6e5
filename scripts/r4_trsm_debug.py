"""round 4 debugging aid: the posterior's blocked solves through nngp_model_apply_factor, size by size, with progress on stderr."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel

for n, rows in [(8192, 512), (8192, 1000), (9216, 512), (9300, 512), (10800, 600), (12288, 256)]:
    x, y = synth.synthetic_queries(n, 24, seed=51)
    model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double()
    B = torch.randn((rows, n), device="cuda", dtype=torch.float32)
    for both in (False, True):
        print("n", n, "rows", rows, "both", both, "...", file=sys.stderr, flush=True)
        X = model.apply_factor(B.clone(), both_halves=both)
        torch.cuda.synchronize()
        Xd = X.double()
        R = (Xd @ L) @ L.T - B.double() if both else Xd @ L.T - B.double()
        print("   residual max", float(R.abs().max()), "x max", float(Xd.abs().max()), file=sys.stderr, flush=True)
    model.close()
print("done")
