"""The serving-mode case of tests/test_gpu_api.py (N = 2500, d = 20) and the ill-conditioned sweep case (N = 2706, d = 3, diag_reg 1e-4)
through the float64 residual product (key 5 = 50) and the int8 one with 5 x 5 / 6 x 6 planes (keys 54 / 52 + 54 is implied by N >= 2048)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
for (n, d, reg, get) in ((2500, 20, 1e-3, "nngp"), (2706, 3, 1e-4, "nngp"), (5000, 20, 1e-3, "nngp"), (3000, 16, 1e-3, "ntk"), (2907, 2, 1e-4, "ntk"), (8192, 64, 1e-3, "ntk")):
    x, y = synth.synthetic_queries(n + 40, d, seed=6)
    xt, _ = synth.synthetic_queries(150, d, seed=106)
    arch = ([0.96] * 4, [0.05] * 4) if d == 3 else ([1.0] * 3, [0.0] * 3)
    model = GPModel(n + 40, d, arch[0], arch[1], get=get, diag_reg=reg, knobs=True).fit(x[:n], y[:n])
    model.debug_set(5, 50)
    model.set_refine(3)
    _, var3 = model.predict(xt, cov="diag")
    res = {"N": n, "d": d, "reg": reg, "get": get, "cg_iters": model.info()["refine_iters"], "var_min": float(var3.min()), "var_max": float(var3.max())}
    for level in (1, 2):
        model.set_refine(level)
        for name, key in (("f64", 50), ("i8_5x5", 0), ("i8_6x6", 52)):
            model.debug_set(5, key)
            _, v = model.predict(xt, cov="diag")
            res["L%d_%s" % (level, name)] = float(np.max(np.abs(v - var3) / np.abs(var3)))
    if get == "ntk":
        print(json.dumps(res), flush=True); model.close(); continue
    model.debug_set(5, 50)
    model.prepare_serving()
    model.set_refine(2)
    for name, key in (("f64", 50), ("i8_5x5", 0), ("i8_6x6", 52)):
        model.debug_set(5, key)
        _, v = model.predict(xt, cov="diag")
        res["serving_%s" % name] = float(np.max(np.abs(v - var3) / np.abs(var3)))
    print(json.dumps(res), flush=True)
    model.close()
