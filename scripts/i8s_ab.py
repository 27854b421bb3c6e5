"""A/B of the int8 residual path at a bench config (knobs library): fit / predict times and the variances with every product on the
float64 pipe (key 5 = 50), the default (COARSE int8 first residual, FINE int8 later products), the coarse product only (key 5 = 57) and
the planes of K cut in stream order instead of beside the first solves (key 5 = 53).  Earlier forms of this
script also timed WHERE the planes of K are cut (beside the Cholesky on a priority stream, late in it, on CU-masked streams):
profiles/r3_i8s_slicing_placement.json."""
import sys, json, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import os
os.environ["NNGP_KNOBS"] = "1"
from nngp_src_amd import _lib, synth
from nngp_src_amd.model import GPModel
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N, d, M, relu, get = {"cfg2": (8192, 64, 1024, 1, "nngp"), "cfg3": (32768, 128, 1024, 3, "nngp"), "cfg5": (16384, 256, 1024, 2, "ntk"),
                      "forest": (10800, 20, 3600, 1, "nngp")}[cfg]
lib = _lib.load()
x, y = synth.synthetic_queries(N, d, seed=0); xt, _ = synth.synthetic_queries(M, d, seed=1)
m = GPModel(N, d, [1.0] * (relu + 1), [0.0] * (relu + 1), get=get, m_cap=M, knobs=True)
xd, yd, xtd = (_lib.to_device_f64(v) for v in (x, y, xt))
def step():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit(xd, yd); torch.cuda.synchronize(); t1 = time.perf_counter()
    mean, var = m.predict(xtd, "diag", as_numpy=False); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, mean, var
out = {"config": cfg}
ref = None
for name, key in (("f64", 50), ("i8s_default", 0), ("i8s_coarse_first_residual_only", 57), ("i8s_planes_in_stream_order", 53)):
    lib.nngp_debug_set(5, key)
    for _ in range(2): step()
    ts = [step() for _ in range(4)]
    out[name] = {"fit_ms": min(t[0] for t in ts), "predict_ms": min(t[1] for t in ts)}
    mean, var = ts[-1][2], ts[-1][3]
    mean = mean.cpu().numpy() if hasattr(mean, "cpu") else np.asarray(mean); var = var.cpu().numpy() if hasattr(var, "cpu") else np.asarray(var)
    if ref is None: ref = (mean, var)
    out[name]["var_rel_vs_f64"] = float(np.max(np.abs(var - ref[1]) / np.abs(ref[1])))
    out[name]["mean_rel_vs_f64"] = float(np.max(np.abs(mean - ref[0]) / np.maximum(1.0, np.abs(ref[0]))))
print(json.dumps(out))
