#!/usr/bin/env python3
"""A/B of the k_build_mfma template variants (knob 5 = 20 + v) at N = 32768, d = 128, n_relu = 3; with ablation masks."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel
n, d, n_relu = 32768, 128, 3
x, y = synth.synthetic_queries(n, d, seed=0)
m = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, knobs=True)
m.set_train(x, y)
names = {1: "KC16 nopf wg4 ldsout", 2: "KC32 pf wg3 direct", 3: "KC32 pf wg3 ldsout", 4: "KC32 pf wg4 direct", 5: "KC16 pf wg4 direct",
         6: "KC16 pf wg3 direct", 7: "KC32 nopf wg4 direct"}
out = {}
for v in range(1, 8):
    m.debug_set(5, 20 + v)
    row = {}
    for mask in (0, 1, 4, 5, 7):
        m.debug_set(3, 32 + mask if mask else 0)
        m.build_rows(0, n); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            m.build_rows(0, n)
        e1.record(); torch.cuda.synchronize()
        row["full" if mask == 0 else "mask%d" % mask] = round(e0.elapsed_time(e1) / 4, 3)
    out[names[v]] = row
    print(names[v], row, flush=True)
m.debug_set(3, 0); m.debug_set(5, 0)
