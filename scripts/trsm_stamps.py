import os; os.environ.setdefault("NNGP_KNOBS", "1")
import sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d = 8192, 64
x, y = synth.synthetic_queries(n, d, seed=0)
m = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, knobs=True)
m.fit(x, y); torch.cuda.synchronize()
m.debug_set(7, 9)
m.set_train(x, y); m.build_rows(0, n); m.factor(); torch.cuda.synchronize()
m.debug_set(7, 0)
