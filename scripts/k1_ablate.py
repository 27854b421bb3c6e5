#!/usr/bin/env python3
"""Timing ablations of k_build_mfma at N = 32768 (knob 3 = 32 + mask: 1 no stores, 2 one k-chunk only, 4 no layer map)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel
n, d, n_relu = 32768, 128, 3
x, y = synth.synthetic_queries(n, d, seed=0)
m = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, knobs=True)
m.set_train(x, y)
out = {}
for mask in (0, 1, 2, 4, 3, 5, 6, 7):
    m.debug_set(3, 32 + mask if mask else 0)
    m.build_rows(0, n); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.build_rows(0, n)
    e1.record(); torch.cuda.synchronize()
    out["mask%d(%s)" % (mask, "+".join(t for b, t in ((1, "nostore"), (2, "nogram"), (4, "nomap")) if mask & b) or "full")] = round(e0.elapsed_time(e1) / 5, 3)
m.debug_set(3, 0)
print(json.dumps(out))
