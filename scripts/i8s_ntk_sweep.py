"""NTK fits at N = 2100 .. 9000 (where the int8 products are used), random architectures / encodings / regularisers: the default path
(COARSE first residual; FINE later products from N >= 4096, M >= 512) against the same predict with every product on the float64 pipe
(timing-knob key 5 = 50) -- variances, the sweep estimate and the decision to continue the rows by CG."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
worst = 0.0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    r = np.random.default_rng(7000 + seed)
    n = int(r.integers(2100, 9000)); m = int(r.choice([300, 600, 1000])); d = int(r.choice([2, 3, 7, 20, 64, 128]))
    n_relu = int(r.integers(1, 4)); w = float(r.uniform(0.7, 1.6)); b = float(r.choice([0.0, 0.05, 0.3])); reg = float(r.choice([1e-4, 1e-3, 1e-2]))
    x, y = synth.synthetic_queries(n, d, seed=seed); xt, _ = synth.synthetic_queries(m, d, seed=100 + seed)
    model = GPModel(n, d, [w] * (n_relu + 1), [b] * (n_relu + 1), get="ntk", diag_reg=reg, m_cap=m, knobs=True).fit(x, y)
    out = {}
    for name, key in (("f64", 50), ("default", 0)):
        model.debug_set(5, key)
        mean, var = model.predict(xt, cov="diag")
        out[name] = (np.asarray(mean), np.asarray(var), model.cov_iters(), model.sweep_estimate()[1])
    model.debug_set(5, 50); model.set_refine(6); _, ref = model.predict(xt, cov="diag"); model.debug_set(5, 0)
    rel = lambda v: float(np.max(np.abs(v - ref) / np.abs(ref)))
    row = dict(seed=seed, n=n, m=m, d=d, n_relu=n_relu, diag_reg=reg, cg_iters=model.info()["refine_iters"],
               err_f64=rel(out["f64"][1]), err_default=rel(out["default"][1]), cov_iters_f64=out["f64"][2], cov_iters_default=out["default"][2],
               est_f64=out["f64"][3], est_default=out["default"][3],
               mean_diff=float(np.max(np.abs(out["default"][0] - out["f64"][0]) / np.maximum(1.0, np.abs(out["f64"][0])))))
    worst = max(worst, row["err_default"] / max(row["err_f64"], 1e-9))
    print(json.dumps(row), flush=True)
    model.close()
print(json.dumps({"worst_err_default_over_err_f64": worst}))
