"""Variance error against level 3 (float64 products) of levels 1 / 2 and of the serving mode, NNGP and NTK, easy and ill-conditioned fits:
all products on the float64 pipe (key 5 = 50), the default (COARSE int8 product for the first residual, FINE 7 x 7 planes for later
residuals and the NTK's W where N >= 4096 and M >= 512), and the coarse product only (key 5 = 57: later products on the float64 pipe)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
for (n, d, reg, get) in ((2500, 20, 1e-3, "nngp"), (2706, 3, 1e-4, "nngp"), (5000, 20, 1e-3, "nngp"), (3000, 16, 1e-3, "ntk"), (2907, 2, 1e-4, "ntk"), (8192, 64, 1e-3, "ntk"), (5007, 2, 1e-4, "ntk"), (8192, 64, 1e-3, "nngp")):
    x, y = synth.synthetic_queries(n + 40, d, seed=6)
    xt, _ = synth.synthetic_queries(600, d, seed=106)
    arch = ([0.96] * 4, [0.05] * 4) if d == 3 else ([1.0] * 3, [0.0] * 3)
    model = GPModel(n + 40, d, arch[0], arch[1], get=get, diag_reg=reg, knobs=True).fit(x[:n], y[:n])
    model.debug_set(5, 50)
    model.set_refine(3)
    _, var3 = model.predict(xt, cov="diag")
    res = {"N": n, "d": d, "reg": reg, "get": get, "cg_iters": model.info()["refine_iters"], "var_min": float(var3.min()), "var_max": float(var3.max())}
    for level in (1, 2):
        model.set_refine(level)
        for name, key in (("f64", 50), ("i8_default", 0), ("i8_coarse_first_only", 57)):
            model.debug_set(5, key)
            _, v = model.predict(xt, cov="diag")
            res["L%d_%s" % (level, name)] = float(np.max(np.abs(v - var3) / np.abs(var3)))
    if get == "ntk":
        print(json.dumps(res), flush=True); model.close(); continue
    model.debug_set(5, 50)
    model.prepare_serving()
    model.set_refine(2)
    for name, key in (("f64", 50), ("i8_default", 0), ("i8_coarse_first_only", 57)):
        model.debug_set(5, key)
        _, v = model.predict(xt, cov="diag")
        res["serving_%s" % name] = float(np.max(np.abs(v - var3) / np.abs(var3)))
    print(json.dumps(res), flush=True)
    model.close()
