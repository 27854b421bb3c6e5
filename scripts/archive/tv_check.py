"""Stand-alone check of the persistent triangular sweeps of the alpha CG (csrc/trsv_tickets.hip): nngp_model_precond against scipy on the
model's own float32 factor, and the time of the full alpha solve.  scripts/tv_check.py N [d]"""
import ctypes, faulthandler, os, sys, time
faulthandler.enable(); faulthandler.dump_traceback_later(200, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel

def P(*a): print(*a, flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9300
d = int(sys.argv[2]) if len(sys.argv) > 2 else 24
x, y = synth.synthetic_queries(n, d, seed=51)
model = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3)
model.set_train(x, y); model.build_rows(0, n); model.factor()
torch.cuda.synchronize()
lib, h, s, p = model.lib, model.handle, _lib.stream_ptr(), _lib.ptr
rng = np.random.default_rng(1)
r = torch.from_numpy(rng.standard_normal(n)).cuda()
z = torch.empty_like(r)
_lib.check(lib.nngp_model_precond(h, p(r), p(z), s), lib)
torch.cuda.synchronize()
z1 = z.cpu().numpy().copy()
_lib.check(lib.nngp_model_precond(h, p(r), p(z), s), lib)
torch.cuda.synchronize()
P("bitwise repeat:", bool(np.array_equal(z1, z.cpu().numpy())), "nan:", int(np.isnan(z1).sum()))
if n <= 13000:
    import scipy.linalg as sla
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double().cpu().numpy()
    zref = sla.cho_solve((L, True), r.cpu().numpy())
    P("precond vs scipy: rel l2", float(np.linalg.norm(z1 - zref) / np.linalg.norm(zref)))
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.nngp_model_precond(h, p(r), p(z), s), lib)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
P("precond (two sweeps + conversions) ms:", ["%.3f" % t for t in ts])
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.solve(); a = model.alpha()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
P("alpha solve ms:", ["%.2f" % t for t in ts], model.info())
model.close()
