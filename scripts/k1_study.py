#!/usr/bin/env python3
"""Where the kernel-build time goes: symmetric N x N build timed for several (d, n_relu) on the GPU box."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import synth, _lib
from nngp_src_amd.model import GPModel

n = 32768
out = {}
for variant in (0, 4):  # 0: k_build_mfma (default); 4: the round-1 all-VALU kernel
    _lib.load(knobs=True).nngp_debug_set(3, variant)
    for d in (16, 128, 256):
        x, y = synth.synthetic_queries(n, d, seed=0)
        for n_relu in (0, 1, 3):
            m = GPModel(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, knobs=True)
            m.set_train(x, y)
            m.build_rows(0, n); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                m.build_rows(0, n)
            e1.record(); torch.cuda.synchronize()
            out["%s_d%d_relu%d_ms" % ("valu_r1" if variant == 4 else "mfma", d, n_relu)] = round(e0.elapsed_time(e1) / 3, 3)
            m.close(); del m
_lib.load(knobs=True).nngp_debug_set(3, 0)
print(json.dumps(out, indent=1))
