#!/usr/bin/env python3
"""End-to-end serving latency of the reference's Estimator surface (estimator.py:42-67): query lines in, (mean, std) out --
native line encoder + H2D copy + predict in serving mode -- on a toy two-table schema with N training queries."""
import contextlib, io, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import encoder as enc
from nngp_src_amd.estimator import Estimator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
rng = np.random.default_rng(0)
lines = []
for i in range(n):
    up, lo = sorted(rng.uniform(0, 100, 2), reverse=True)
    wu, wl = sorted(rng.uniform(-5, 5, 2), reverse=True)
    ku, kl = sorted(rng.uniform(0, 10, 2), reverse=True)
    card = max(1, int(5000 * (up - lo) / 100 * (wu - wl) / 10 * (ku - kl) / 10))
    lines.append("a@k,%.3f,%.3f#v,%.2f,%.2f#w,%.2f,%.2f@@%d" % (ku, kl, up, lo, wu, wl, card))
tables = [enc.TableEncoder("a", [enc.numerical("k", 0, 10), enc.numerical("v", 0, 100), enc.numerical("w", -5, 5)], 64),
          enc.TableEncoder("b", [enc.numerical("k", 0, 10), enc.categorical("c", 70)], 64)]
out = {"N": n}
with tempfile.TemporaryDirectory() as tmp:
    open(os.path.join(tmp, "q.txt"), "w").write("\n".join(lines) + "\n")
    for serving in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            est = Estimator("toy", "", tmp, encoder=enc.NNGPEncoder(tables), serving=serving)
            t0 = time.perf_counter(); est.load_model(); est.predict([lines[0].rsplit("@", 1)[0]])
            out["load_model_s_%s" % ("serving" if serving else "solve")] = round(time.perf_counter() - t0, 3)
            for m in (1, 8, 64, 512):
                q = [l.rsplit("@", 1)[0] for l in lines[:m]]
                est.predict(q)
                t0 = time.perf_counter()
                for _ in range(10):
                    est.predict(q)
                out["predict_%d_lines_ms_%s" % (m, "serving" if serving else "solve")] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
print(json.dumps(out, indent=1))
