"""Kernel-only timing of the split-float16 GEMM per tile-order variant (torch events around the launch; the split and
the allocation are outside the timed region because the entry point is called once to warm up)."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load()
dev = G.dev()
torch.manual_seed(0)
for (m, k, lower) in [(30720, 1024, True), (16384, 1024, True), (8192, 1024, True), (16384, 1024, False)]:
    a = torch.randn((m, k), device=dev)
    c = torch.zeros((m, m), device=dev)
    for staging in (0,):
        lib.nngp_debug_set(0, staging)
        for variant in range(4):
            lib.nngp_debug_set(5, 10 + variant)
            for rep in range(2):
                G.gemm_nt_h3(c, a, a, -1.0, 1.0, 2.0 ** 10, lower)
    del a, c
