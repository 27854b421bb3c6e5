#!/usr/bin/env python3
"""bench.py -- NNGP kernel-build + GP-solve on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one synthetic batch of encoded queries already
resident in HBM: row norms + regulariser, N x N kernel build, float32 MFMA Cholesky, CG solve for alpha
on the float64 kernel, and the posterior (cross kernel, mean, diag variance) for M test queries.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg4|cfg5|cfg1]

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL).  "Strong" scaling: the problem is fixed.
Default multi-rank mode ("replicate"): every rank runs the fit (build + Cholesky + alpha) on its own GPU and the test
rows are sharded -- the float64 kernel is built at ~0.9 TB/s on one GPU, faster than xGMI can move it, so no
data-path collective pays at these sizes.  NNGP_DIST_MODE=shard selects the north-star layout
instead: row-block kernel shard + one all-gather + block-cyclic Cholesky with one broadcast per block column.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {
    # name: (N, d, n_relu, get, M, join_block, description)  -- BASELINE.json configs, SURVEY.md 8d
    "cfg1": (1000, 20, 1, "nngp", 200, False, "forest-like N=1000/M=200, d=20, 3-layer ReLU NNGP (configs[0] shape, synthetic rows)"),
    "cfg2": (8192, 64, 1, "nngp", 1024, False, "synthetic N=8192, d=64, 3-layer ReLU NNGP (configs[1])"),
    "cfg3": (32768, 128, 3, "nngp", 1024, False, "synthetic N=32768, d=128, 5-layer ReLU NNGP + full Cholesky posterior (configs[2], north-star target)"),
    "cfg4": (65536, 128, 3, "nngp", 1024, False, "synthetic N=65536, d=128, row-block kernel shard + all-gather (configs[3])"),
    "cfg5": (16384, 256, 1, "ntk", 1024, True, "synthetic join encoding N=16384, d=256, NTK (configs[4]; mean only)"),
}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak (the roofline BASELINE.json's north star names)
PEAK_F16_MFMA_TFLOPS = 2516.6  # dense f16/bf16 MFMA peak = 16 x the f32 one; the split-float16 GEMM spends 3 products per term


def flop_model(n, d, m, n_relu):
    """Algorithmic work of one step."""
    f_k = n * (n + 1) * d
    f_c = n ** 3 / 3 + n ** 2 / 2 + n / 6
    f_solve = 2 * n * n
    f_post = 2 * n * m * d + 2 * n * m + n * n * m + 2 * n * m
    return {"kernel_build": f_k, "cholesky": f_c, "alpha_solve": f_solve, "posterior": f_post,
            "total": f_k + f_c + f_solve + f_post, "relu_maps": n_relu * n * (n + 1) // 2}


def cpu_baseline(n_relu, get):
    """The C float64/OpenMP oracle ("port") on a bounded sample of the same workload, host cores of this box."""
    n, d, m = 6144, 128, 256
    try:  # second opinion on the dominant stage, timed BEFORE the OpenMP oracle spins up its threads:
        import scipy.linalg  # LAPACK dpotrf through SciPy on an SPD matrix of the same size
        g = np.random.default_rng(0).standard_normal((n, 256))
        a = g @ g.T / 256 + np.eye(n)
        t0 = time.perf_counter()
        scipy.linalg.cho_factor(a, lower=True, overwrite_a=True, check_finite=False)
        tl = time.perf_counter() - t0
        lapack = {"ms": round(tl * 1e3, 1), "gflops": round((n ** 3 / 3) / tl / 1e9, 1),
                  "note": "scipy.linalg.cho_factor float64 on a random SPD matrix, N=%d" % n}
        del a, g
    except Exception as e:  # pragma: no cover
        lapack = {"error": str(e)}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from nngp_src_amd import synth
    c_oracle.set_threads(min(16, os.cpu_count() or 1))  # the GPU box gives one GPU a 16-core CPU share
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(m, d, seed=1)
    w, b = [1.0] * (n_relu + 1), [0.0] * (n_relu + 1)
    c_oracle.kernel_build(x[:256], None, "nngp", w, b)  # warm the thread pool
    t0 = time.perf_counter()
    model = c_oracle.fit(x, y, w, b, get="nngp")
    c_oracle.predict_nngp(model, xt, 1)
    dt = time.perf_counter() - t0
    fl = flop_model(n, d, m, n_relu)
    out = {"value": round(fl["total"] / dt / 1e9, 3), "unit": "GFLOP/s", "cores": c_oracle.num_threads(), "kind": "port",
           "sample": "same step at N=%d, d=%d, M=%d, n_relu=%d, float64 C/OpenMP oracle: %.2f s (build %.2f, potrf %.2f)"
                     % (n, d, m, n_relu, dt, model["stage_sec"][0], model["stage_sec"][1]),
           "ms": round(dt * 1e3, 1)}
    out["lapack_dpotrf"] = lapack
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nngp_src_amd import distributed, synth
    from nngp_src_amd.model import GPModel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        # NNGP_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share
        # devices, collectives staged through the host); the real runs use nccl = RCCL over xGMI.
        backend = os.environ.get("NNGP_DIST_BACKEND", "nccl")
        local_dev = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    knobs = bool(os.environ.get("NNGP_DEBUG"))
    if knobs:  # timing experiments only (scripts/gpu_ab.sh), e.g. NNGP_DEBUG="2=1" disables the look-ahead Cholesky: runs on
        from nngp_src_amd import _lib  # libnngp_hip_knobs.so -- the product library has no such switches
        for kv in os.environ["NNGP_DEBUG"].split(","):
            k, v = kv.split("=")
            _lib.load(knobs=True).nngp_debug_set(int(k), int(v))
    n, d, n_relu, get, m, join_block, desc = CONFIGS[args.config]
    x, y = synth.synthetic_queries(n, d, seed=0, join_block=join_block)
    xt, _ = synth.synthetic_queries(m, d, seed=1, join_block=join_block)
    xd, yd, xtd = (torch.from_numpy(a).to(dev) for a in (x, y, xt))
    m0, m1 = distributed.row_partition(m, world, rank)
    xt_local = xtd[m0:m1].contiguous()
    w_std, b_std = [1.0] * (n_relu + 1), [0.0] * (n_relu + 1)
    n_cap = distributed.row_chunk(n, world) * world
    model = GPModel(n_cap, d, w_std, b_std, get=get, diag_reg=1e-3, m_cap=max(m1 - m0, 1), knobs=knobs)
    if os.environ.get("NNGP_REFINE"):  # covariance precision level (default: the library's)
        model.set_refine(int(os.environ["NNGP_REFINE"]))
    cov = "diag" if get == "nngp" else False
    shard = world > 1 and os.environ.get("NNGP_DIST_MODE", "replicate") == "shard"

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def step(stages=None):
        e0 = ev()
        model.set_train(xd, yd)
        e1 = ev()
        r0, r1 = distributed.row_partition(n, world, rank)
        model.build_rows(r0, r1) if shard else model.build_rows(0, n)
        e2 = ev()
        if shard:
            buf, _ = model.kernel_buffer(all_rows=True)
            distributed.allgather_rows(buf, n)
        e3 = ev()
        if shard and os.environ.get("NNGP_DIST_CHOL", "1") != "0":
            distributed.distributed_factor(model)  # block columns dealt cyclically, one broadcast per column
        else:
            model.factor()
        e4 = ev()
        model.solve()
        e5 = ev()
        out = model.predict(xt_local, cov=cov, as_numpy=False) if m1 > m0 else None
        e6 = ev()
        if stages is not None:
            torch.cuda.synchronize()
            for k, (a, b) in {"set_train": (e0, e1), "kernel_build": (e1, e2), "allgather": (e2, e3),
                              "cholesky": (e3, e4), "alpha_solve": (e4, e5), "posterior": (e5, e6)}.items():
                stages.setdefault(k, []).append(a.elapsed_time(b))
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    stages = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(stages)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    info = model.info()
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.config)
    if os.path.exists(tpath):  # HBM bytes of the Cholesky kernels of one step (rocprofv3 --pmc, scripts/gpu_pmc.sh)
        traffic = json.load(open(tpath)).get("cholesky_bytes")
        traffic_src = "profiles/pmc_traffic_%s.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)" % args.config
    if rank == 0:
        fl = flop_model(n, d, m, n_relu)
        ms = elapsed / args.steps * 1e3
        st = {k: float(np.mean(v)) for k, v in stages.items()}
        chol_tflops = fl["cholesky"] / (st["cholesky"] * 1e-3) / 1e12
        result = {
            "metric": "NNGP kernel-build + GP-solve wall-clock (ms) and GFLOP/s at N train queries",
            "value": round(fl["total"] / (ms * 1e-3) / 1e9, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (Cholesky products as split f16 x3, f32 accumulate; f64 build/CG/means)", "data": "synthetic",
            "config": {"workload": desc, "N": n, "d": d, "n_relu": n_relu, "get": get, "M_test": m,
                       "parallelism": ("single GPU" if world == 1 else
                                       ("row-block kernel shard x%d + all-gather, block-cyclic Cholesky (broadcast per block column), "
                                        "replicated solve, test rows sharded" % world) if shard else
                                       "fit replicated on %d GPUs (no data-path collective), test rows sharded" % world),
                       "precision": "float64 kernel build + CG residual; float32 Cholesky (preconditioner) whose trailing updates run "
                                    "as split-float16 MFMA products (hi+lo, 3 per term, float32 accumulate); float64 means"},
            # Cholesky stage = the dominant cost.  `achieved` = algorithmic F_C / stage time.  Its matrix work runs on the
            # float16 pipe at 3 products per float32-grade term, so the hardware peak for it is PEAK_F16 / 3; the
            # fraction of the float32-MFMA roofline the north star names is reported beside it (it can exceed 1).
            "roofline": {"bound": "mfma", "achieved": round(chol_tflops, 3), "peak": round(PEAK_F16_MFMA_TFLOPS / 3, 1),
                         "unit": "TFLOP/s", "frac": round(chol_tflops / (PEAK_F16_MFMA_TFLOPS / 3), 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "Cholesky stage (k_gemm_nt_h3 trailing updates + k_gemm_nt_f32 panel GEMMs + k_potrf_leaf), "
                                   "F_C = N^3/3 + N^2/2 + N/6 per step",
                         "peak_note": "dense f16 MFMA peak %.1f TF/s / 3 products per term; executed MFMA flops = 3 x achieved"
                                      % PEAK_F16_MFMA_TFLOPS,
                         "f32_mfma_peak": PEAK_F32_MFMA_TFLOPS,
                         "frac_of_f32_mfma_peak": round(chol_tflops / PEAK_F32_MFMA_TFLOPS, 4),
                         "north_star_frac_build_plus_cholesky": round((fl["kernel_build"] + fl["cholesky"]) /
                                                                       ((st["kernel_build"] + st["cholesky"]) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
            "stages_ms": {k: round(v, 3) for k, v in st.items()},
            "stages_note": "the CG solve for alpha is deferred to the posterior stage, where it runs on its own stream under the "
                           "covariance products (alpha_solve only records the request); with a covariance it stops at 1e-6 and "
                           "the mean is corrected through the covariance rows (mu = K_td a_k + Z r_k, ~1e-9 of the converged "
                           "mean); fit_info is read after the timed steps, where info() runs the solve on to 1e-10",
            "fit_info": {"cg_iters": info["refine_iters"], "rel_residual": info["rel_residual"],
                         "clamped_pivots": info["clamped_pivots"], "reg": info["reg"],
                         "alpha_l2": float(torch.linalg.vector_norm(model.alpha()).item())},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(n_relu, get)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
