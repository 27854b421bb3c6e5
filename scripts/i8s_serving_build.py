import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
for (n, d, reg) in ((5000, 20, 1e-3), (8192, 64, 1e-3), (8100, 3, 1e-4)):
    x, y = synth.synthetic_queries(n, d, seed=6); xt, _ = synth.synthetic_queries(600, d, seed=106)
    arch = ([0.96] * 4, [0.05] * 4) if d == 3 else ([1.0] * 3, [0.0] * 3)
    model = GPModel(n, d, arch[0], arch[1], diag_reg=reg, knobs=True).fit(x, y)
    model.debug_set(5, 50); model.set_refine(3); _, var3 = model.predict(xt, cov="diag")
    res = {"N": n, "d": d}
    for name, key in (("f64_built", 50), ("fine_built", 0)):
        model.debug_set(5, key)
        torch.cuda.synchronize(); t0 = time.perf_counter(); model.prepare_serving(); torch.cuda.synchronize(); res[name + "_build_ms"] = (time.perf_counter() - t0) * 1e3
        model.debug_set(5, 50); model.set_refine(2)
        _, v = model.predict(xt, cov="diag"); res[name + "_serving_err"] = float(np.max(np.abs(v - var3) / np.abs(var3)))
        model.fit(x, y)
    print(json.dumps(res), flush=True); model.close()
