#!/usr/bin/env python3
"""GPU micro-benchmarks of the building blocks (run on the GPU box): GEMM shapes, Cholesky leaf, potrf sizes."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")  # timing-knob build of the library
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from nngp_src_amd import _lib  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lib = _lib.load(knobs=True)
    dev = G.dev()
    out = {}
    torch.manual_seed(0)
    for (m, n, k, lower) in [(16384, 16384, 4096, False), (16384, 16384, 4096, True), (8192, 8192, 8192, False),
                             (4096, 4096, 4096, False), (2048, 2048, 2048, False), (1024, 1024, 1024, False),
                             (512, 512, 512, False), (256, 256, 256, False), (128, 128, 128, False),
                             (16384, 128, 128, False), (1024, 128, 128, False)]:
        a = torch.randn((m, k), device=dev); b = torch.randn((n, k), device=dev); c = torch.zeros((m, n), device=dev)

        def run():
            _lib.check(lib.nngp_gemm_nt_f32(_lib.ptr(c), n, _lib.ptr(a), k, _lib.ptr(b), k, m, n, k, -1.0, 1.0, int(lower),
                                            _lib.stream_ptr()))
        ms = timeit(run, reps=3 if m >= 8192 else 20)
        fl = (m * (m + 128) * k) if lower else 2.0 * m * n * k
        out["gemm_%dx%dx%d%s" % (m, n, k, "_lower" if lower else "")] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2)}
        del a, b, c
    for n in (128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        base = torch.randn((n, n), device=dev)
        spd = base @ base.T / n + torch.eye(n, device=dev) * 2.0
        a = spd.clone()
        dinv = torch.empty((n // 128, 128, 128), device=dev)
        clamped = torch.zeros(1, dtype=torch.int32, device=dev)

        def run():
            a.copy_(spd)
            _lib.check(lib.nngp_potrf_f32(_lib.ptr(a), n, n, _lib.ptr(dinv), _lib.ptr(clamped), _lib.stream_ptr()))
        ms = timeit(run, reps=10 if n <= 2048 else 3)
        cp = timeit(lambda: a.copy_(spd), reps=10)
        ms -= cp
        out["potrf_%d" % n] = {"ms": round(ms, 4), "tflops": round(n ** 3 / 3 / ms / 1e9, 2), "clamped": int(clamped.item())}
        del base, spd, a
    # leaf ablations (dbg bit 1: skip diagonal factor/inverse, 2: skip inverse assembly, 4: skip MFMA sub-block updates)
    base = torch.randn((128, 128), device=dev)
    spd = base @ base.T / 128 + torch.eye(128, device=dev) * 2.0
    a = spd.clone(); dinv = torch.empty((1, 128, 128), device=dev); clamped = torch.zeros(1, dtype=torch.int32, device=dev)
    for variant in (0, 1):
        lib.nngp_debug_set(3, variant)
        for dbg in (0, 1, 2, 4, 7):
            lib.nngp_debug_set(0, dbg)
            ms = timeit(lambda: _lib.check(lib.nngp_potrf_f32(_lib.ptr(a), 128, 128, _lib.ptr(dinv), _lib.ptr(clamped), _lib.stream_ptr())), reps=50)
            out["leaf_v%d_dbg%d_us" % (variant, dbg)] = round(ms * 1e3, 2)
    lib.nngp_debug_set(0, 0)
    lib.nngp_debug_set(3, 0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
