import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
for (n, d, m, relu) in ((32768, 128, 1024, 3), (8192, 64, 1024, 1), (10800, 20, 3600, 1)):
    x, y = synth.synthetic_queries(n, d, seed=0); xt, _ = synth.synthetic_queries(m, d, seed=1)
    xd, yd, xtd = (torch.from_numpy(a).cuda() for a in (x, y, xt))
    model = GPModel(n, d, [1.0] * (relu + 1), [0.0] * (relu + 1), diag_reg=1e-3, m_cap=m, knobs=True)
    res = {"N": n}
    for e in (6, 0, 4, 2):   # 6: stop at 1e-6 only (key 3 = 60 would do the same); 0: the default (early stop + third-order correction)
        model.debug_set(3, 60 if e == 6 else (0 if e == 0 else 40 + e))
        ts = []
        for rep in range(4):
            model.set_train(xd, yd); model.build_rows(0, n); model.factor(); model.solve()
            torch.cuda.synchronize(); t0 = time.perf_counter(); mean, var = model.predict(xtd, cov="diag", as_numpy=False); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res["tol_%s_ms" % ("1e-6_only" if e == 6 else "default_early_stop" if e == 0 else "1e-%d" % e)] = round(min(ts), 2)
        mm = mean.cpu().numpy()
        if e == 6: ref = mm
        else: res["tol_%s_mean_diff" % ("default_early_stop" if e == 0 else "1e-%d" % e)] = float(np.max(np.abs(mm - ref) / np.maximum(1.0, np.abs(ref))))
    model.debug_set(3, 0)
    print(json.dumps(res), flush=True); model.close()
