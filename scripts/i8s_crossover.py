"""Where the int8 residual path pays: predict (diag variance, level 1) with the float64 residual product (timing-knob key 5 = 50)
against the int8 plane products, over training sizes N and blocks of M test rows; the planes of K are cut once per fit, so the
first predict after a fit (which pays the slicing) and a later one are both timed."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel
d = 64
out = []
for n in (2048, 4096, 8192, 16384):
    x, y = synth.synthetic_queries(n, d, seed=0)
    xd, yd = _lib.to_device_f64(x), _lib.to_device_f64(y)
    model = GPModel(n, d, [1.0] * 3, [0.0] * 3, diag_reg=1e-3, m_cap=2048, knobs=True)
    # the library's own threshold is bypassed: key 5 = 54 forces the int8 path at any size
    for m in (128, 256, 512, 1024, 2048):
        xt, _ = synth.synthetic_queries(m, d, seed=1)
        xtd = _lib.to_device_f64(xt)
        row = {"N": n, "M": m}
        for name, key in (("f64", 50), ("i8s", 54)):
            model.debug_set(5, key)
            first, later = [], []
            for rep in range(3):
                model.fit(xd, yd); torch.cuda.synchronize()
                t0 = time.perf_counter(); mean, var = model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize(); first.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter(); mean, var = model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize(); later.append((time.perf_counter() - t0) * 1e3)
            row[name + "_first_ms"] = round(min(first), 3); row[name + "_later_ms"] = round(min(later), 3)
            v = var.cpu().numpy()
            if name == "f64": ref = v
            else: row["var_rel_diff"] = float(np.max(np.abs(v - ref) / np.abs(ref)))
        out.append(row); print(json.dumps(row), flush=True)
    model.close()
