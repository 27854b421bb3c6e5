"""Stand-alone check of the persistent blocked solves (csrc/trsm_tickets.hip) against scipy, with progress lines and a
watchdog: scripts/tk_check.py N ROWS [REPS].  Prints the residual gates of tests/test_gpu_parity.py::
test_blocked_solves_of_the_posterior_against_scipy and the time per solve (forward / both halves)."""
import faulthandler
import os
import sys
import time

faulthandler.enable()
faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from nngp_src_amd import synth
from nngp_src_amd.model import GPModel


def P(*a):
    print(*a, flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 9300
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    knobs = os.environ.get("NNGP_KNOBS", "0") == "1"
    dev = torch.device("cuda", 0)
    x, y = synth.synthetic_queries(n, 24, seed=51)
    P("fit", n)
    model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, knobs=knobs).fit(x, y)
    torch.cuda.synchronize()
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double().cpu().numpy()
    rng = np.random.default_rng(n + rows)
    B = rng.standard_normal((rows, n)).astype(np.float32) * (1.0 + 10.0 * rng.random((rows, 1)).astype(np.float32))
    P("forward solve")
    X = model.apply_factor(torch.from_numpy(B.copy()).to(dev))
    torch.cuda.synchronize()
    P("forward solve done; info:", model.info())
    X = X.cpu().numpy().astype(np.float64)
    P("nan:", int(np.isnan(X).sum()))
    if os.environ.get("TK_MAP", "0") == "1":
        import scipy.linalg as sla
        Xref = sla.solve_triangular(L, B.astype(np.float64).T, lower=True).T
        npad = -(-n // 1024) * 1024
        P("per (row tile, block column): log10 of max |X - Xref| / max |Xref| of the block, 'N' = NaN")
        for r in range(0, rows, 128):
            line = []
            for J in range(0, n, 1024):
                blk = X[r:r + 128, J:J + 1024]; ref = Xref[r:r + 128, J:J + 1024]
                if np.isnan(blk).any():
                    line.append("  N%3d" % (100 * np.isnan(blk).mean()))
                else:
                    line.append("%5.1f" % np.log10(max(np.abs(blk - ref).max() / np.abs(ref).max(), 1e-99)))
            P("r%2d " % (r // 128) + " ".join(line))
        if os.environ.get("TK_ROWS", "0") == "1":
            nanmask = np.isnan(X)
            for J in range(0, min(n, 3072), 1024):
                blk = nanmask[:, J:J + 1024]
                rows_bad = np.where(blk.any(axis=1))[0]
                P("block", J // 1024, "rows with NaN:", rows_bad.tolist()[:80])
                for rr in rows_bad[:6]:
                    cols = np.where(blk[rr])[0]
                    P("   row", int(rr), "NaN cols: count", len(cols), "first", cols[:4].tolist(), "last", cols[-4:].tolist())
    if n <= 13000:
        scale = np.abs(X) @ np.abs(L).T
        res = np.abs(X @ L.T - B)
        P("forward: max res/scale", float((res / scale).max()))
    P("both halves")
    Z = model.apply_factor(torch.from_numpy(B.copy()).to(dev), both_halves=True)
    torch.cuda.synchronize()
    Z = Z.cpu().numpy().astype(np.float64)
    P("nan:", int(np.isnan(Z).sum()), "info:", model.info())
    if n <= 13000:
        Y = Z @ L
        scale2 = np.abs(Y) @ np.abs(L).T + (np.abs(Z) @ np.abs(L)) @ np.abs(L).T
        res2 = np.abs(Y @ L.T - B)
        P("both: max res/scale", float((res2 / scale2).max()))
    # run-to-run bits
    Z2 = model.apply_factor(torch.from_numpy(B.copy()).to(dev), both_halves=True).cpu().numpy().astype(np.float64)
    P("bitwise repeat:", bool(np.array_equal(Z, Z2)))
    bt = torch.from_numpy(B.copy()).to(dev)
    for mode, name in ((False, "forward"), (True, "both halves")):
        ts = []
        for _ in range(reps):
            b = bt.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.apply_factor(b, both_halves=mode)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        P("%s: ms per call (incl. copies)" % name, ["%.3f" % t for t in ts])
    model.close()


if __name__ == "__main__":
    main()
