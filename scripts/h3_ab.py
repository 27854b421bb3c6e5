"""A/B of split-float16 GEMM variants on the Cholesky's update shape and the blocked solves' shape, kernel times from a
rocprofv3 kernel trace of this script (scripts/gpu_h3_ab.sh): variants are run in interleaved rounds, the trace is
matched to them by dispatch order."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load(knobs=True)
torch.manual_seed(0)
VARIANTS = [int(v) for v in os.environ.get("H3_VARIANTS", "0").split(",")]
KEY = int(os.environ.get("H3_KEY", "5"))
SHAPES = [(30720, 30720, 1024, True), (16384, 16384, 1024, True), (1024, 31744, 1024, False)]
plan = []
for (m, n, k, lower) in SHAPES:
    a = torch.randn((m, k), device=G.dev())
    b = a if lower else torch.randn((n, k), device=G.dev())
    c = torch.zeros((m, n), device=G.dev())
    for rnd in range(6):
        for v in VARIANTS:
            _lib.check(lib.nngp_debug_set(KEY, v))
            _lib.check(lib.nngp_gemm_nt_h3(_lib.ptr(c), c.stride(0), _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), m, n, k,
                                           -1.0, 1.0, 2.0 ** 10, int(lower), _lib.stream_ptr()))
            torch.cuda.synchronize()
            plan.append({"shape": [m, n, k, lower], "variant": v, "round": rnd})
    del a, b, c
json.dump(plan, open(os.path.join(ROOT, "gpurun_out", "h3_ab_plan.json"), "w"))
