#!/usr/bin/env python3
"""bench.py -- NNGP kernel-build + GP-solve on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one synthetic batch of encoded queries already
resident in HBM: row norms + regulariser, N x N kernel build, float32 MFMA Cholesky, CG solve for alpha
on the float64 kernel, and the posterior (cross kernel, mean, diag variance) for M test queries.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg4|cfg5|cfg1] [--mode shard|shard32|replicate|grid2d]

N = 1: configs[2] (cfg3, N = 32768, the north-star size).  N > 1 (launched by torch.distributed.run, one rank per GPU):
configs[3] (cfg4, N = 65536) in the layout BASELINE.json's north star names -- rank g builds the row block
[g N/G, (g+1) N/G) of the float64 kernel, ONE in-place RCCL all-gather over xGMI completes K on every rank
(nngp_allgather_rows of the C ABI; torch.distributed's nccl all_gather_into_tensor if the library's own communicator
cannot be created), then the 1-D block-cyclic Cholesky (one broadcast per block column), replicated alpha solve, test
rows sharded.  "Strong" scaling: the problem is fixed.  The line also carries the three numbers SURVEY.md 8e asks for
-- sharded build, all-gather, replicated full build -- and, measured after the timed steps in the same invocation,
the step time of the "replicate" layout (every rank runs the whole fit, no data-path collective) for comparison.
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
    "cfg5": (16384, 256, 1, "ntk", 1024, True, "synthetic join encoding N=16384, d=256, NTK mean + NTK-ensemble variance (configs[4])"),
}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak (the roofline BASELINE.json's north star names)
PEAK_F16_MFMA_TFLOPS = 2516.6  # dense f16/bf16 MFMA peak = 16 x the f32 one; the split-float16 GEMM spends 3 products per term
PEAK_I8_MFMA_TOPS = 5033.2     # dense int8 MFMA peak = 2 x the f16 one (MI355X_MICROARCH.md, matrix-core table)
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _i8_bytes_per_launch(n_pad, m_pad, pairs):
    """Algorithmic bytes of one k_gemm_nt_i8s launch: digit planes of both operands read once + int32 plane products written once."""
    def one(nz, nk, ndiag, chunk):
        return float(nk) * n_pad * n_pad + float(nz) * m_pad * n_pad + 4.0 * ndiag * m_pad * n_pad * max(1, -(-n_pad // chunk))
    coarse, fine = one(3, 5, 5, 40960), one(5, 7, 7, 16384)
    f = min(1.0, max(0.0, (pairs - 12.0) / 13.0))
    return (1.0 - f) * coarse + f * fine


def flop_model(n, d, m, n_relu):
    """Algorithmic work of one step."""
    f_k = n * (n + 1) * d
    f_c = n ** 3 / 3 + n ** 2 / 2 + n / 6
    f_solve = 2 * n * n
    f_post = 2 * n * m * d + 2 * n * m + n * n * m + 2 * n * m
    return {"kernel_build": f_k, "cholesky": f_c, "alpha_solve": f_solve, "posterior": f_post,
            "total": f_k + f_c + f_solve + f_post, "relu_maps": n_relu * n * (n + 1) // 2}


def cpu_baseline(n_bench, d, m, n_relu, gpu_ms):
    """The host build of the same C ABI (oracle/libnngp_cpu.so: include/nngp_hip.h on the float64 C/OpenMP oracle, "port") on
    bounded samples of the same workload, host cores of this box: N in {4096, 8192, 16384}, least-squares
    fit t = a N^2 + b N^3, labelled extrapolation to N_bench."""
    sys.path.insert(0, ROOT)
    from oracle import c_abi
    from nngp_src_amd import synth
    lapack = None
    try:  # second opinion on the dominant stage, timed BEFORE the OpenMP oracle spins up its threads:
        import scipy.linalg  # LAPACK dpotrf through SciPy on an SPD matrix
        nl = 6144
        g = np.random.default_rng(0).standard_normal((nl, 256))
        a = g @ g.T / 256 + np.eye(nl)
        t0 = time.perf_counter()
        scipy.linalg.cho_factor(a, lower=True, overwrite_a=True, check_finite=False)
        tl = time.perf_counter() - t0
        lapack = {"ms": round(tl * 1e3, 1), "gflops": round((nl ** 3 / 3) / tl / 1e9, 1),
                  "note": "scipy.linalg.cho_factor float64 on a random SPD matrix, N=%d" % nl}
        del a, g
    except Exception as e:  # pragma: no cover
        lapack = {"error": str(e)}
    c_abi.set_threads(min(16, os.cpu_count() or 1))  # the GPU box gives one GPU a 16-core CPU share
    w, b = [1.0] * (n_relu + 1), [0.0] * (n_relu + 1)
    sizes = [int(v) for v in os.environ.get("NNGP_CPU_SIZES", "4096,8192,16384").split(",")]
    mm = m  # the GPU step's own number of test queries (round 2 timed 256 on the CPU against 1024 on the GPU)
    xt, _ = synth.synthetic_queries(mm, d, seed=1)
    c_abi.kernel_build(xt, None, "nngp", w, b)  # warm the thread pool
    samples = []
    for n in sizes:
        x, y = synth.synthetic_queries(n, d, seed=0)
        model = c_abi.CpuModel(n, d, w, b, get="nngp", diag_reg=1e-3, m_cap=mm)
        t0 = time.perf_counter()
        model.set_train(x, y)
        model.build_rows(0, n)
        t1 = time.perf_counter()
        model.factor()
        t2 = time.perf_counter()
        model.solve()
        model.predict(xt, "diag")
        dt = time.perf_counter() - t0
        fl = flop_model(n, d, mm, n_relu)
        samples.append({"N": n, "M": mm, "sec": round(dt, 3), "build_sec": round(t1 - t0, 3),
                        "potrf_sec": round(t2 - t1, 3), "gflops": round(fl["total"] / dt / 1e9, 2)})
        del model, x, y
    ns = np.array([s["N"] for s in samples], dtype=np.float64)
    ts = np.array([s["sec"] for s in samples])
    coef, *_ = np.linalg.lstsq(np.stack([ns ** 2, ns ** 3], axis=1), ts, rcond=None)
    t_bench = float(coef[0] * n_bench ** 2 + coef[1] * n_bench ** 3)
    big = samples[-1]
    out = {"value": big["gflops"], "unit": "GFLOP/s", "cores": c_abi.num_threads(), "kind": "port",
           "sample": "same step (set_train + build_rows + factor + solve + predict(diag) for M=%d) through the host build of the same "
                     "C ABI (oracle/libnngp_cpu.so: float64 C/OpenMP) at N = %s, d=%d, n_relu=%d; `value` is the largest sample"
                     % (mm, sizes, d, n_relu),
           "samples": samples,
           "extrapolation": {"model": "t = a N^2 + b N^3 (least squares over the samples)", "a": float(coef[0]), "b": float(coef[1]),
                             "N": n_bench, "cpu_sec_extrapolated": round(t_bench, 2),
                             "gpu_ms_per_step": round(gpu_ms, 3),
                             "speedup_vs_extrapolated_cpu": round(t_bench * 1e3 / gpu_ms, 1),
                             "note": "EXTRAPOLATED, not measured at N=%d (same M = %d test queries as the GPU step)" % (n_bench, mm)},
           "lapack_dpotrf": lapack}
    return out


def bench_grid2d(args, world, rank, dev, cfg_name, x, y, xt, n, d, n_relu, get, m, desc):
    """Opt-in comparison layout: the 2-D block-cyclic fit of nngp-src_amd/dist2d.py.  A step = build of the
    rank's own tiles + distributed Cholesky + distributed CG for alpha + mean and level-1 variance of the test queries (replicated
    right-hand-side blocks of 128).  Correctness-first code: one trailing update per block column (split-float16 from the second column on), torch glue between kernels."""
    import torch
    import torch.distributed as dist
    from nngp_src_amd import dist2d
    pr = int(os.environ.get("NNGP_GRID_ROWS", "0")) or max(p for p in range(1, int(world ** 0.5) + 1) if world % p == 0)
    pc = world // pr
    grid = dist2d.Grid(pr, pc)
    ops = dist2d.HipOps([1.0] * (n_relu + 1), [0.0] * (n_relu + 1), get=get)
    mt = min(m, int(os.environ.get("NNGP_GRID_TEST_ROWS", "128")))

    def step():
        gp = dist2d.Dist2DGP(ops, grid, x, y, diag_reg=1e-3, nb=1024).fit()
        gp.predict(xt[:mt])
        return gp

    for _ in range(args.warmup):
        step()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gp = step()
    dist.barrier(); torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        fl = flop_model(n, d, mt, n_relu)
        ms = float(t.item()) / args.steps * 1e3
        print(json.dumps({
            "metric": "NNGP kernel-build + GP-solve wall-clock (ms) and GFLOP/s at N train queries",
            "value": round(fl["total"] / (ms * 1e-3) / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (tile Cholesky; trailing updates as split f16 x3 from the second block column on; f64 build/CG/means)", "data": "synthetic",
            "config": {"workload": desc, "N": n, "d": d, "n_relu": n_relu, "get": get, "M_test": mt,
                       "parallelism": "2-D block-cyclic %d x %d (tiles of 1024): no rank holds the whole kernel or factor; panel broadcasts "
                                      "along process rows, exchanges inside process columns, fan-in triangular solves" % (pr, pc)},
            "fit_info": {"cg_iters": gp.cg_iters, "rel_residual": gp.relres, "clamped_pivots": gp.clamped, "reg": gp.reg,
                         "alpha_l2": float(torch.linalg.vector_norm(gp.alpha).item())},
            "shard": {"ranks_seen": dist.get_world_size(), "local_tile_bytes": int(gp.a32.numel() * 4 + gp.k64t.numel() * 8)},
            "roofline": None, "cpu_baseline": None}), flush=True)
    dist.destroy_process_group()


def bench_shard32(args, world, rank, dev, cfg_name, x, y, xt, n, d, n_relu, get, m, desc):
    """Opt-in layout (SURVEY.md 8e with the exchange it sizes; nngp-src_amd/shard32.py): every rank keeps its float64 kernel rows, ONE
    all-gather carries the float32 factor input, the Cholesky is replicated (NNGP_DIST_CHOL=1: 1-D block-cyclic), the alpha CG's
    matrix-vector product and the covariance's residual product are sharded by the rows a rank holds.  A step = sharded kernel build +
    float32 all-gather + factorisation + CG + mean and level-1 variance of the M test queries (dealt to the ranks)."""
    import torch
    import torch.distributed as dist
    from nngp_src_amd import shard32
    if get != "nngp":
        raise SystemExit("bench.py --mode shard32: the row-sharded covariance is the NNGP one")
    ops = shard32.HipRowOps(n, d, [1.0] * (n_relu + 1), [0.0] * (n_relu + 1), diag_reg=1e-3, world=world)
    dchol = os.environ.get("NNGP_DIST_CHOL", "0") == "1"

    def step():
        gp = shard32.RowShardedGP(ops, x, y, distributed_cholesky=dchol).fit()
        gp.predict(xt, cov=True)
        return gp

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gp = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        fl = flop_model(n, d, m, n_relu)
        ms = float(t.item()) / args.steps * 1e3
        chunk = -(-n // world)
        print(json.dumps({
            "metric": "NNGP kernel-build + GP-solve wall-clock (ms) and GFLOP/s at N train queries",
            "value": round(fl["total"] / (ms * 1e-3) / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (Cholesky products as split f16 x3, f32 accumulate; f64 build/CG/means; float64 residual product on the float64 MFMA)",
            "data": "synthetic",
            "config": {"workload": desc, "N": n, "d": d, "n_relu": n_relu, "get": get, "M_test": m,
                       "parallelism": "row-block kernel shard x%d, float64 rows kept local; ONE in-place all-gather of the float32 factor input; %s "
                                      "Cholesky; CG matrix-vector product and residual product sharded by rows (one N-vector / M x N/G all-gather per use)"
                                      % (world, "1-D block-cyclic" if dchol else "replicated")},
            "fit_info": {"cg_iters": gp.cg_iters, "rel_residual": gp.relres, "reg": gp.reg, "shift_scale": gp.shift_scale,
                         "alpha_l2": float(torch.linalg.vector_norm(gp.alpha).item())},
            "shard": {"ranks_seen": world, "layout": "shard32",
                      "allgather_GB_received_per_rank": round(gp.exchanged_bytes["factor_input_received_per_rank"] / 1e9, 4),
                      "float64_kernel_allgather_GB_would_be": round((world - 1) * chunk * ops.factor_input_buffer().shape[1] * 8 / 1e9, 4),
                      "cg_vector_GB_received_per_rank": round(gp.exchanged_bytes["cg_vectors_received_per_rank"] / 1e9, 6),
                      "backend": dist.get_backend() if world > 1 else "none"},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher around it: start the ranks as a CHILD `torch.distributed.run` of this same
    file with the same arguments, relay rank 0's JSON line and exit with the child's code.  Runs before this process has imported
    torch or touched the GPU (a process that has initialised HIP must never exec or be replaced; a child process is fine)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    limit = float(os.environ.get("NNGP_BENCH_TIMEOUT", "3000"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, err = child.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)  # the process group this launcher started, nothing else
        except ProcessLookupError:
            pass
        out, err = child.communicate()
        sys.stderr.write(err[-4000:])
        sys.stderr.write("\nbench.py: the %d-rank child run exceeded %.0f s and was killed\n" % (n_ranks, limit))
        sys.exit(124)
    line = None
    for ln in out.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    if (child.returncode != 0 or line is None) and env.get("NNGP_COLLECTIVE", "native") == "native" and "--mode" not in sys.argv:
        # one more attempt with every collective through torch.distributed's own RCCL process group instead of the library's
        # communicator (a different configuration, named in the line) -- a crash there must not cost the whole measurement
        sys.stderr.write(err[-3000:])
        sys.stderr.write("\nbench.py: child run failed (code %d); second attempt with NNGP_COLLECTIVE=torch\n" % child.returncode)
        first_rc, first_err = child.returncode, err[-1500:]
        env["NNGP_COLLECTIVE"] = "torch"
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            cmd[cmd.index("--master-port") + 1] = str(s.getsockname()[1])
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        try:
            out, err = child.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.communicate()
            sys.exit(124)
        line = None
        for ln in out.splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                d = json.loads(ln)
                d["launcher_note"] = "first attempt (library-owned RCCL communicator) failed; this line is the NNGP_COLLECTIVE=torch run"
                d["first_attempt"] = {"returncode": first_rc, "stderr_tail": first_err}  # so that gating sees the failure behind rc 0
                line = json.dumps(d)
    if child.returncode != 0 or line is None:
        sys.stderr.write(err[-6000:])
        sys.stderr.write("\nbench.py: the %d-rank child run (%s) ended with code %d%s\n"
                         % (n_ranks, " ".join(cmd[1:4]), child.returncode, "" if line else " and printed no result line"))
        sys.exit(child.returncode or 1)
    print(line, flush=True)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: cfg3 on one GPU, cfg4 on several")
    ap.add_argument("--mode", default=os.environ.get("NNGP_DIST_MODE", "shard"), choices=["shard", "replicate", "grid2d", "shard32"],
                    help="multi-GPU layout (default: the north star's row-block shard + all-gather; grid2d: the 2-D block-cyclic "
                         "fit of dist2d.py, no rank holding the whole kernel or factor)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compare", action="store_true", help="multi-GPU: skip the untimed comparison legs")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)  # no torch import, no GPU call before this point

    import torch
    import torch.distributed as dist
    from nngp_src_amd import _lib, distributed, synth
    from nngp_src_amd.model import GPModel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            sys.exit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
        # NNGP_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share
        # devices, collectives staged through the host); the real runs use nccl = RCCL over xGMI.  RCCL refuses two ranks
        # on one device, so with fewer devices than ranks the rehearsal backend is taken (and named in the line).
        n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if n_dev < 1:
            sys.exit("bench.py: no GPU visible")
        backend = os.environ.get("NNGP_DIST_BACKEND") or ("nccl" if n_dev >= world else "gloo")
        local_dev = local_rank % n_dev
        torch.cuda.set_device(local_dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    ranks_seen = dist.get_world_size() if world > 1 else 1
    cfg_name = args.config or ("cfg3" if world == 1 else "cfg4")

    knobs = bool(os.environ.get("NNGP_DEBUG"))
    if knobs:  # timing experiments only (scripts/gpu_ab.sh), e.g. NNGP_DEBUG="2=1" disables the look-ahead Cholesky: runs on
        for kv in os.environ["NNGP_DEBUG"].split(","):  # libnngp_hip_knobs.so -- the product library has no such switches
            k, v = kv.split("=")
            _lib.load(knobs=True).nngp_debug_set(int(k), int(v))
    n, d, n_relu, get, m, join_block, desc = CONFIGS[cfg_name]
    x, y = synth.synthetic_queries(n, d, seed=0, join_block=join_block)
    xt, _ = synth.synthetic_queries(m, d, seed=1, join_block=join_block)
    if world > 1 and args.mode == "grid2d":
        return bench_grid2d(args, world, rank, dev, cfg_name, x, y, xt, n, d, n_relu, get, m, desc)
    if args.mode == "shard32" and (world > 1 or os.environ.get("NNGP_SHARD32_SINGLE") == "1"):
        return bench_shard32(args, world, rank, dev, cfg_name, x, y, xt, n, d, n_relu, get, m, desc)
    xd, yd, xtd = (torch.from_numpy(a).to(dev) for a in (x, y, xt))
    m0, m1 = distributed.row_partition(m, world, rank)
    xt_local = xtd[m0:m1].contiguous()
    w_std, b_std = [1.0] * (n_relu + 1), [0.0] * (n_relu + 1)
    n_cap = distributed.row_chunk(n, world) * world
    model = GPModel(n_cap, d, w_std, b_std, get=get, diag_reg=1e-3, m_cap=max(m1 - m0, 1), knobs=knobs)
    if os.environ.get("NNGP_REFINE"):  # covariance precision level (default: the library's)
        model.set_refine(int(os.environ["NNGP_REFINE"]))
    cov = "diag"  # the reference asks for the covariance and consumes its diagonal (train.py:157-158,180), for nngp and ntk
    shard = world > 1 and args.mode == "shard"
    dist_chol = os.environ.get("NNGP_DIST_CHOL", "1") != "0"

    # the shard's one data-path collective: the library's own RCCL communicator (C ABI), torch.distributed otherwise
    comm, collective = None, None
    if shard:
        collective = "torch.distributed %s all_gather_into_tensor (in place)" % backend
        if backend == "nccl" and os.environ.get("NNGP_COLLECTIVE", "native") == "native":
            try:
                comm = distributed.NativeComm()
                collective = "libnngp_hip nngp_allgather_rows: ncclAllGather in place (%s)" % comm.library
            except _lib.NngpError as e:  # NativeComm's rendezvous is failure-symmetric: it raises on every rank or on none
                collective += " [native RCCL communicator unavailable: %s]" % str(e)[:120]

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def step(stages=None, sharded=shard):
        e0 = ev()
        model.set_train(xd, yd)
        e1 = ev()
        r0, r1 = distributed.row_partition(n, world, rank)
        model.build_rows(r0, r1) if sharded else model.build_rows(0, n)
        e2 = ev()
        if sharded:
            buf, _ = model.kernel_buffer(all_rows=True)
            distributed.allgather_rows(buf, n, comm=comm)
        e3 = ev()
        if sharded and dist_chol:
            distributed.distributed_factor(model, comm=comm)  # block columns dealt cyclically, one broadcast per column (nngp_bcast with the library's communicator)
        else:
            model.factor()
        e4 = ev()
        model.solve()
        e5 = ev()
        out = model.predict(xt_local, cov=cov, as_numpy=False) if m1 > m0 else None
        e6 = ev()
        if stages is not None:
            torch.cuda.synchronize()
            if ktimer["on"]:  # HIP events the library put around every trailing-update launch of this step's factorisation
                nl, tms, tfl = model.update_timer_read()
                ktimer["launches"] += nl; ktimer["ms"] += tms; ktimer["flops"] += tfl; ktimer["bytes"] += model.update_timer_bytes()
            if rtimer["on"]:  # ... and around every int8 plane-product launch of this step's predict
                nl, tms, tfl, tops = model.residual_timer_read()
                rtimer["launches"] += nl; rtimer["ms"] += tms; rtimer["flops"] += tfl; rtimer["ops"] += tops
            if stimer["on"]:  # ... and around every blocked triangular solve of this step's predict
                nl, tms, tfl = model.trsm_timer_read()
                stimer["solves"] += nl; stimer["ms"] += tms; stimer["flops"] += tfl
            for k, (a, b) in {"set_train": (e0, e1), "kernel_build": (e1, e2), "allgather": (e2, e3),
                              "cholesky": (e3, e4), "alpha_solve": (e4, e5), "posterior": (e5, e6)}.items():
                stages.setdefault(k, []).append(a.elapsed_time(b))
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(values):
        if world == 1:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    # live timing of the dominant kernel (k_gemm_nt_h3, the Cholesky's split-float16 trailing update) for `roofline`
    ktimer = {"on": False, "launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}
    if not (shard and dist_chol):
        try:
            model.update_timer(True)
            ktimer["on"] = True
        except _lib.NngpError:
            pass
    # ... and of the posterior's residual products on the int8 pipe (k_gemm_nt_i8s) for `roofline_residual`
    rtimer = {"on": False, "launches": 0, "ms": 0.0, "flops": 0.0, "ops": 0.0}
    try:
        model.residual_timer(True)
        rtimer["on"] = True
    except _lib.NngpError:
        pass
    # ... and of the posterior's blocked triangular solves for `roofline_solves`
    stimer = {"on": False, "solves": 0, "ms": 0.0, "flops": 0.0}
    try:
        model.trsm_timer(True)
        stimer["on"] = True
    except _lib.NngpError:
        pass
    for _ in range(args.warmup):
        step()
    barrier()
    if rtimer["on"]:
        model.residual_timer_read()  # drop the warm-up launches
    if stimer["on"]:
        model.trsm_timer_read()
    stages = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(stages)
    barrier()
    elapsed = time.perf_counter() - t0
    (elapsed,) = max_over_ranks([elapsed])

    info = model.info()
    alpha_l2 = float(torch.linalg.vector_norm(model.alpha()).item())
    names = sorted(stages)
    st = dict(zip(names, max_over_ranks([float(np.mean(stages[k])) for k in names])))  # slowest rank per stage

    # ---- comparison legs, after the timed region (multi-GPU): full build on every rank, and the replicate layout ----
    shard_report = None
    if world > 1 and shard:
        chunk = distributed.row_chunk(n, world)
        ld = model.kernel_buffer()[1]
        recv_bytes = (world - 1) * chunk * ld * 8
        shard_report = {"sharded_build_ms": round(st["kernel_build"], 3), "allgather_ms": round(st["allgather"], 3),
                        "allgather_GB_received_per_rank": round(recv_bytes / 1e9, 3),
                        "allgather_GBps_per_rank": round(recv_bytes / 1e9 / (st["allgather"] * 1e-3), 1) if st["allgather"] > 0 else None,
                        "replicated_full_build_ms": None, "replicate_layout_ms_per_step": None,
                        "ranks_seen": ranks_seen, "collective": collective,
                        "cholesky": ("1-D block-cyclic, one broadcast per 1024-wide block column (%s)" % ("nngp_bcast on a side stream" if comm is not None and os.environ.get("NNGP_BCAST", "torch") == "native" else "torch.distributed broadcast")) if dist_chol else "replicated on every rank"}
        if not args.no_compare:
            model.set_train(xd, yd)
            ts = []
            for _ in range(3):
                barrier()
                a_ = ev(); model.build_rows(0, n); b_ = ev()
                torch.cuda.synchronize()
                ts.append(a_.elapsed_time(b_))
            (full_ms,) = max_over_ranks([min(ts)])
            shard_report["replicated_full_build_ms"] = round(full_ms, 3)
            step(sharded=False)
            barrier()
            tr = time.perf_counter()
            for _ in range(2):
                step(sharded=False)
            barrier()
            (rep,) = max_over_ranks([(time.perf_counter() - tr) / 2 * 1e3])
            shard_report["replicate_layout_ms_per_step"] = round(rep, 3)

    traffic, traffic_src, k_traffic, r_traffic, t_traffic = None, None, None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % cfg_name)
    traffic_note = None
    if os.path.exists(tpath):  # HBM bytes of the Cholesky kernels of one step (rocprofv3 --pmc, scripts/gpu_pmc.sh)
        _pm = json.load(open(tpath))
        # the counters are only quoted while the kernel sources they were taken with are unchanged (scripts/pmc_traffic.py records
        # their hashes): a stale file would silently describe kernels that no longer run
        import hashlib
        _csrc = os.path.join(ROOT, "nngp-src_amd", "csrc")
        _sha = _pm.get("sources_sha16")
        _changed = sorted(f for f, h in (_sha or {}).items()
                          if not os.path.exists(os.path.join(_csrc, f)) or hashlib.sha256(open(os.path.join(_csrc, f), "rb").read()).hexdigest()[:16] != h)
        if _sha is None or _changed:
            traffic_note = ("profiles/pmc_traffic_%s.json %s: traffic not reported" % (cfg_name, "carries no source fingerprint" if _sha is None
                            else "predates a change of " + ", ".join(_changed)))
            _pm = {}
        traffic = _pm.get("cholesky_bytes")
        _k = _pm.get("kernels", {}).get("k_gemm_nt_h3v2<true>") or _pm.get("kernels", {}).get("k_gemm_nt_h3<true>")
        if _k and _k.get("calls"):
            k_traffic = (_k["fetch_bytes"] + _k["write_bytes"]) / _k["calls"]
        traffic_src = traffic_note or "profiles/pmc_traffic_%s.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)" % cfg_name
        _r = _pm.get("kernels", {}).get("k_gemm_nt_i8s")
        r_traffic = (_r["fetch_bytes"] + _r["write_bytes"]) / _r["calls"] if _r and _r.get("calls") else None
        _t = _pm.get("kernels", {}).get("k_trsm_tickets")
        t_traffic = (_t["fetch_bytes"] + _t["write_bytes"]) / _t["calls"] if _t and _t.get("calls") else None
    if rank == 0:
        fl = flop_model(n, d, m, n_relu)
        ms = elapsed / args.steps * 1e3
        chol_tflops = fl["cholesky"] / (st["cholesky"] * 1e-3) / 1e12
        post_ms = st["posterior"] + st["alpha_solve"]
        post_tflops = fl["posterior"] / (post_ms * 1e-3) / 1e12
        k_bytes = (8 * n * n + 8 * n * d) * (1 if world == 1 or not shard else 1.0 / world)
        k_gbps = k_bytes / (st["kernel_build"] * 1e-3) / 1e9
        result = {
            "metric": "NNGP kernel-build + GP-solve wall-clock (ms) and GFLOP/s at N train queries",
            "value": round(fl["total"] / (ms * 1e-3) / 1e9, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (Cholesky products as split f16 x3, f32 accumulate; f64 build/CG/means; the posterior's float64-grade residual as exact int8 digit-plane products)", "data": "synthetic",
            "config": {"workload": desc, "N": n, "d": d, "n_relu": n_relu, "get": get, "M_test": m,
                       "parallelism": ("single GPU" if world == 1 else
                                       ("row-block kernel shard x%d + one in-place all-gather, %s, replicated alpha solve, test rows sharded"
                                        % (world, "block-cyclic Cholesky (broadcast per block column)" if dist_chol else "replicated Cholesky")) if shard else
                                       "fit replicated on %d GPUs (no data-path collective), test rows sharded" % world),
                       "precision": "float64 kernel build + CG residual; float32 Cholesky (preconditioner) whose trailing updates run "
                                    "as split-float16 MFMA products (hi+lo, 3 per term, float32 accumulate); float64 means; variances: "
                                    "float32-grade solves on the float16 pipe + one float64-GRADE residual product on the int8 pipe "
                                    "(operands cut into exact 8-bit digit planes, 12 exact plane products, float64 combination; level 1)"},
            "roofline": None,  # filled in below: the dominant kernel when the library timed it, else the stage
            # Cholesky stage = the dominant cost.  `achieved` = algorithmic F_C / stage time.  Its matrix work runs on the
            # float16 pipe at 3 products per float32-grade term, so the hardware peak for it is PEAK_F16 / 3; the
            # fraction of the float32-MFMA roofline the north star names is reported beside it (it can exceed 1).
            "roofline_cholesky_stage": {"bound": "mfma", "achieved": round(chol_tflops, 3), "peak": round(PEAK_F16_MFMA_TFLOPS / 3, 1),
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
            # the other two stages against the rooflines SURVEY.md 8d names for them
            "roofline_posterior": {"bound": "mfma", "achieved": round(post_tflops, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(post_tflops / PEAK_F32_MFMA_TFLOPS, 4), "stage_ms": round(post_ms, 3),
                                   "work": "2NMd + 4NM + N^2 M (cross kernel, mean, variances) for this rank's %d test rows x world" % (m1 - m0)
                                           if world > 1 else "2NMd + 4NM + N^2 M (cross kernel, mean, variances)",
                                   "note": "algorithmic count; executed: one float64-grade residual product 2 N^2 M as 12 exact int8 plane "
                                           "products (k_gemm_nt_i8s, see roofline_residual; until round 3 one float64 MFMA product on the "
                                           "78.6 TF/s pipe), three float32-grade triangular solves N^2 M each on the float16 pipe, the "
                                           "alpha CG on HBM"},
            "roofline_k1": {"bound": "hbm", "achieved": round(k_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                            "frac": round(k_gbps / PEAK_HBM_GBPS, 4), "stage_ms": round(st["kernel_build"], 3),
                            "bytes": "8 N^2 + 8 N d (float64 K written once, X read once)" + ("" if world == 1 or not shard else " / world")},
            "stages_ms": {k: round(v, 3) for k, v in st.items()},
            "stages_note": "the CG solve for alpha is deferred to the posterior stage, where it runs on its own stream under the "
                           "covariance products (alpha_solve only records the request); with a covariance it stops at 1e-6 and "
                           "the mean is corrected through the covariance rows (mu = K_td a_k + Z r_k, ~1e-9 of the converged "
                           "mean); fit_info is read after the timed steps, where info() runs the solve on to 1e-10"
                           + ("; stages are the slowest rank's" if world > 1 else ""),
            "fit_info": {"cg_iters": info["refine_iters"], "rel_residual": info["rel_residual"],
                         "clamped_pivots": info["clamped_pivots"], "reg": info["reg"], "alpha_l2": alpha_l2},
        }
        if ktimer["on"] and ktimer["launches"] > 0 and ktimer["ms"] > 0.0:
            # `roofline` = the DOMINANT KERNEL, measured live: HIP events around every k_gemm_nt_h3<lower> launch, on the stream
            # it is launched on (nngp_model_update_timer), summed over the timed steps.  achieved = algorithmic flops of the
            # launches (2 x updated entries x panel width: float32-grade work) / their summed duration; the kernel executes
            # three float16 MFMA products per term, so the pipe's peak for this work is PEAK_F16 / 3.
            k_tflops = ktimer["flops"] / (ktimer["ms"] * 1e-3) / 1e12
            per_step = ktimer["launches"] / args.steps
            result["roofline"] = {
                "bound": "mfma", "achieved": round(k_tflops, 3), "peak": round(PEAK_F16_MFMA_TFLOPS / 3, 1), "unit": "TFLOP/s",
                "frac": round(k_tflops / (PEAK_F16_MFMA_TFLOPS / 3), 4),
                "traffic": k_traffic, "traffic_source": (traffic_src + ", k_gemm_nt_h3v2<true>, per launch") if k_traffic else traffic_note,
                "traffic_over_algorithmic": round(k_traffic / (ktimer["bytes"] / ktimer["launches"]), 3) if k_traffic else None,
                "kernel": "k_gemm_nt_h3v2<LOWER=true> (split-float16 updates of the grouped look-ahead Cholesky: K = 1024 inside a group "
                          "of 4 block columns, K = 4096 beyond it)",
                "launches_per_step": round(per_step, 2), "avg_launch_ms": round(ktimer["ms"] / ktimer["launches"], 4),
                "ms_per_step_in_kernel": round(ktimer["ms"] / args.steps, 3),
                "algorithmic_flops_per_launch": round(ktimer["flops"] / ktimer["launches"], 1),
                "algorithmic_bytes_per_launch": round(ktimer["bytes"] / ktimer["launches"], 1),
                "bytes_note": "C read + written once per launch (8 B per updated float32 entry) + the operands' split rows once (4 B per "
                              "row and k): nngp_model_update_timer_bytes",
                "executed_f16_mfma_tflops": round(3 * k_tflops, 1),
                "peak_note": "dense f16 MFMA peak %.1f TF/s / 3 products per float32-grade term" % PEAK_F16_MFMA_TFLOPS,
                "timer": "HIP events on the update stream around each launch (library: nngp_model_update_timer), timed steps only",
            }
        else:
            result["roofline"] = dict(result["roofline_cholesky_stage"])
        if rtimer["on"] and rtimer["launches"] > 0 and rtimer["ms"] > 0.0:
            # the second-largest kernel, measured live like the first: HIP events around every k_gemm_nt_i8s launch on the stream
            # it is launched on.  achieved = int8 operations executed (2 m n k x plane pairs) / summed duration.
            r_tops = rtimer["ops"] / (rtimer["ms"] * 1e-3) / 1e12
            n_pad, m_pad = -(-n // 128) * 128, -(-(m1 - m0) // 128) * 128
            result["roofline_residual"] = {
                "bound": "mfma", "achieved": round(r_tops, 1), "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s (int8)",
                "frac": round(r_tops / PEAK_I8_MFMA_TOPS, 4),
                "kernel": "k_gemm_nt_i8s (the posterior's float64-grade residual products as exact int8 digit-plane products)",
                "launches_per_step": round(rtimer["launches"] / args.steps, 2), "ms_per_step_in_kernel": round(rtimer["ms"] / args.steps, 3),
                "plane_pairs": round(rtimer["ops"] / rtimer["flops"], 2),
                # algorithmic bytes per launch: the digit planes of both operands read once + the int32 plane products written once
                # (a first residual: 5 planes of K, 3 of the rounded z, 5 diagonals in chunks of <= 40960 k -- 12 pairs; a FINE product:
                # 7 planes of K, 5 of the rounded z, 7 diagonals in chunks of <= 16384 -- 25 pairs; a step's mix follows from its mean pair count)
                "algorithmic_bytes_per_launch": round(_i8_bytes_per_launch(n_pad, m_pad, rtimer["ops"] / rtimer["flops"])
                                                      * (rtimer["flops"] / rtimer["launches"]) / (2.0 * m_pad * n_pad * n_pad), 1),
                "traffic": r_traffic, "traffic_source": (traffic_src + ", k_gemm_nt_i8s, per launch") if r_traffic else traffic_note,
                "float64_equivalent_tflops": round(rtimer["flops"] / (rtimer["ms"] * 1e-3) / 1e12, 1),
                "float64_mfma_peak": 78.6,
                "note": "the float64 matrix pipe peaks at 78.6 TF/s (the kernel this replaced ran the same product at 68); the chip holds "
                        "~1.6 GHz under this load (profiles/r3_i8s_pmc_summary.txt: MFMA pipe 73 % busy)",
                "timer": "HIP events on the caller's stream around each launch (library: nngp_model_residual_timer), timed steps only",
            }
        if stimer["on"] and stimer["solves"] > 0 and stimer["ms"] > 0.0:
            # the posterior's blocked triangular solves: one persistent, ticket-ordered launch each since round 5 (csrc/trsm_tickets.hip),
            # HIP events on the caller's stream around each solve
            s_tf = stimer["flops"] / (stimer["ms"] * 1e-3) / 1e12
            n_pad, m_pad = -(-n // 128) * 128, -(-(m1 - m0) // 128) * 128
            result["roofline_solves"] = {
                "bound": "mfma", "achieved": round(s_tf, 1), "peak": round(PEAK_F16_MFMA_TFLOPS / 3.0, 1), "unit": "TFLOP/s",
                "frac": round(s_tf / (PEAK_F16_MFMA_TFLOPS / 3.0), 4),
                "kernel": "k_trsm_tickets: one persistent launch per blocked triangular solve of the posterior (256 x 256 split-float16 update "
                          "tiles, 128 x 128 diagonal-product and chain tiles, row splits, all drawn from a ticket counter), N^2 M "
                          "algorithmic flops per solve",
                "solves_per_step": round(stimer["solves"] / args.steps, 2), "ms_per_step": round(stimer["ms"] / args.steps, 3),
                # algorithmic bytes of one solve: the factor's split copy read once (lower triangle, 4 bytes an entry) + the right-hand
                # sides read and written once; the counters show what the 128 / 256-row tiles actually move through the fabric
                "algorithmic_bytes_per_launch": round(2.0 * n_pad * n_pad + 8.0 * m_pad * n_pad, 1),
                "traffic": t_traffic, "traffic_source": (traffic_src + ", k_trsm_tickets, per launch") if t_traffic else traffic_note,
                "note": "a dependency chain of N / 1024 block columns (DESIGN.md section 4); the solves share the chip with the alpha CG (from "
                        "the predict's start) and the first two with the cut of K's digit planes",
                "timer": "HIP events on the caller's stream around each solve (library: nngp_model_trsm_timer), timed steps only",
            }
        if shard_report is not None:
            result["shard"] = shard_report
        elif world > 1:
            result["shard"] = {"ranks_seen": ranks_seen, "note": "replicate layout: no data-path collective"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(n, d, m, n_relu, ms)
        print(json.dumps(result), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
