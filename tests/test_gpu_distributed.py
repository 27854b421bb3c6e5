"""Two ranks sharing the one GPU of the test box (gloo, host-staged collectives): the multi-GPU fit path --
row-sharded kernel build, all-gather, block-cyclic Cholesky with one broadcast per block column -- must reproduce
the single-rank posterior.  The production runs use backend nccl (RCCL over xGMI) with the same code path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, out_dir, hard=False):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nngp_src_amd import distributed, synth
        from nngp_src_amd.model import GPModel
        x, y = synth.synthetic_queries(n, d, seed=0)
        xt, _ = synth.synthetic_queries(64, d, seed=1)
        n_cap = distributed.row_chunk(n, world) * world
        if hard:  # cond(K + reg I) * eps32 >> 1: the block-cyclic float32 factorisation clamps pivots
            model = GPModel(n_cap, d, [1.63] * 4, [0.0] * 4, diag_reg=10.0, diag_reg_absolute_scale=True)
        else:
            model = GPModel(n_cap, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3)
        distributed.sharded_fit(model, x, y)
        info = model.info()
        assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
        assert (model.factor_shift() > info["reg"]) == hard
        mean, var = model.predict(xt, cov="diag")
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, alpha=model.alpha().cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [2048 + 77])
def test_two_rank_fit_matches_single_rank(tmp_path, n):
    d = 32
    sys.path.insert(0, ROOT)
    from nngp_src_amd import synth
    from nngp_src_amd.model import GPModel
    mp.spawn(_worker, args=(2, _free_port(), n, d, str(tmp_path)), nprocs=2, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(64, d, seed=1)
    ref = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    m0, v0 = ref.predict(xt, cov="diag")
    a0 = ref.alpha().cpu().numpy()
    for r in range(2):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.linalg.norm(g["alpha"] - a0) / np.linalg.norm(a0) < 1e-9
        assert np.linalg.norm(g["mean"] - m0) / np.linalg.norm(m0) < 1e-9
        np.testing.assert_allclose(g["var"], v0, rtol=1e-6)


def test_two_rank_fit_survives_a_float32_breakdown(tmp_path):
    """The ill-conditioned fit of test_gpu_parity (d = 2, 4-layer, absolute diag_reg = 10): the ranks agree that the
    block-cyclic factorisation broke down, refactor with a raised preconditioner shift, and still deliver the
    single-rank posterior."""
    n, d = 3000, 2
    sys.path.insert(0, ROOT)
    from nngp_src_amd import synth
    from nngp_src_amd.model import GPModel
    mp.spawn(_worker, args=(2, _free_port(), n, d, str(tmp_path), True), nprocs=2, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(64, d, seed=1)
    ref = GPModel(n, d, [1.63] * 4, [0.0] * 4, diag_reg=10.0, diag_reg_absolute_scale=True).fit(x, y)
    m0, v0 = ref.predict(xt, cov="diag")
    for r in range(2):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.linalg.norm(g["mean"] - m0) / np.linalg.norm(m0) < 1e-7
        np.testing.assert_allclose(g["var"], v0, rtol=1e-5)


def test_native_rccl_communicator_single_rank():
    """The section-e entry points of the C ABI on the one GPU of the test box: librccl is found and bound at run time,
    a one-rank communicator is created from a unique id, and the in-place all-gather / broadcast run on the caller's
    stream and leave the data as it is (world = 1: every row block is the rank's own).  Two ranks cannot share a
    device under RCCL, so the multi-rank exchange itself is covered by the gloo tests above and the driver's 8-GPU run."""
    sys.path.insert(0, ROOT)
    from nngp_src_amd import distributed, _lib
    torch.cuda.set_device(0)
    comm = distributed.NativeComm()
    assert comm.world == 1 and comm.rank == 0
    assert "rccl" in comm.library.lower() or comm.library.startswith("symbols"), comm.library
    k = torch.arange(6 * 8, dtype=torch.float64, device="cuda").reshape(6, 8).contiguous()
    before = k.clone()
    comm.allgather_rows(k, 5)
    v = torch.arange(33, dtype=torch.float32, device="cuda")
    comm.bcast(v, 0)
    torch.cuda.synchronize()
    assert torch.equal(k, before) and torch.equal(v, torch.arange(33, dtype=torch.float32, device="cuda"))
    lib = _lib.load()
    assert lib.nngp_bcast(_lib.ptr(v), 33, _lib.DTYPE_F32, 3, comm.handle, None) != 0  # root outside the communicator
    # the panel-broadcast plumbing of the block-cyclic Cholesky: nngp_bcast on a side stream, the compute stream waits by event
    nb = distributed._NativeBcast(comm)
    big = torch.randn(1 << 20, dtype=torch.float32, device="cuda")
    keep = big.clone()
    handle = nb.start(big, 0)
    handle.wait()
    out = big * 2.0  # ordered behind the transfer on the compute stream
    torch.cuda.synchronize()
    assert torch.equal(big, keep) and torch.equal(out, keep * 2.0)
    comm.close()


def _worker_rccl(rank, world, port, n, d, out_dir, bcast):
    """One rank per GPU, backend nccl (= RCCL): the library-owned communicator end to end."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["NNGP_BCAST"] = bcast
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from nngp_src_amd import distributed, synth, _lib
        from nngp_src_amd.model import GPModel
        comm = distributed.NativeComm()
        assert comm.world == world and comm.rank == rank
        # nngp_allgather_rows in place: every rank fills its own row block, all of them end up with every block
        rows = 5 * world + 3
        chunk = distributed.row_chunk(rows, world)
        k = torch.full((chunk * world, 16), -1.0, dtype=torch.float64, device="cuda")
        r0, r1 = distributed.row_partition(rows, world, rank)
        k[r0:r1] = float(rank + 1)
        comm.allgather_rows(k, rows)
        torch.cuda.synchronize()
        for q in range(world):
            q0, q1 = distributed.row_partition(rows, world, q)
            assert torch.all(k[q0:q1] == float(q + 1)), (rank, q)
        # nngp_bcast from every root
        for root in range(world):
            v = torch.full((1000,), float(rank), dtype=torch.float32, device="cuda")
            comm.bcast(v, root)
            torch.cuda.synchronize()
            assert torch.all(v == float(root))
        # the fit: row-sharded build, in-place all-gather, block-cyclic factorisation with broadcast panels
        x, y = synth.synthetic_queries(n, d, seed=0)
        xt, _ = synth.synthetic_queries(64, d, seed=1)
        n_cap = distributed.row_chunk(n, world) * world
        model = GPModel(n_cap, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3)
        distributed.sharded_fit(model, x, y, comm=comm)
        info = model.info()
        assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
        mean, var = model.predict(xt, cov="diag")
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, alpha=model.alpha().cpu().numpy())
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bcast", ["torch", "native"])
def test_library_owned_rccl_communicator_on_two_gpus(tmp_path, bcast):
    """SURVEY.md 8e on real links, as soon as a box with two or more GPUs runs this suite (rounds 1-4 only ever had one: skipped
    there): the library's OWN RCCL communicator (csrc/collective.cpp: dlopen'd librccl, unique id carried by torch.distributed) at
    world = 2 -- nngp_allgather_rows in place, nngp_bcast from every root, and distributed.sharded_fit (row-sharded kernel build,
    all-gather, 1-D block-cyclic factorisation with one broadcast per block column) against the single-GPU fit: alpha to 1e-9,
    no clamped pivot.  `native`: the panels through nngp_bcast on a side stream (NNGP_BCAST=native, opt-in until this test has
    passed once on hardware); `torch`: through torch.distributed's broadcast (the default)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: the library-owned RCCL communicator cannot put two ranks on one device")
    n, d = 4096 + 300, 32
    sys.path.insert(0, ROOT)
    from nngp_src_amd import synth
    from nngp_src_amd.model import GPModel
    mp.spawn(_worker_rccl, args=(2, _free_port(), n, d, str(tmp_path), bcast), nprocs=2, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(64, d, seed=1)
    torch.cuda.set_device(0)
    ref = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    m0, v0 = ref.predict(xt, cov="diag")
    a0 = ref.alpha().cpu().numpy()
    for r in range(2):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.linalg.norm(g["alpha"] - a0) / np.linalg.norm(a0) < 1e-9
        assert np.linalg.norm(g["mean"] - m0) / np.linalg.norm(m0) < 1e-9
        np.testing.assert_allclose(g["var"], v0, rtol=1e-6)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_two_gpu_%s.txt" % bcast), "w") as f:
            f.write("library-owned RCCL communicator at world = 2 (NNGP_BCAST=%s): all-gather, broadcast, sharded fit == single-GPU fit\n" % bcast)


def _worker_shard32(rank, world, port, n, m, d, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nngp_src_amd import shard32, synth
        x, y = synth.synthetic_queries(n, d, seed=0)
        xt, _ = synth.synthetic_queries(m, d, seed=1)
        ops = shard32.HipRowOps(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, world=world)
        gp = shard32.RowShardedGP(ops, x, y).fit()
        assert gp.relres <= 1e-10 and gp.cg_iters <= 10 and gp.shift_scale == 1.0, (gp.relres, gp.cg_iters, gp.shift_scale)
        mean, var = gp.predict(xt, cov=True)
        # the model itself serves means from the installed alpha (its float64 kernel holds this rank's rows only)
        m_model = ops.model.predict(xt, cov=False)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, alpha=ops.to_host(gp.alpha), m_model=np.ravel(m_model),
                 recv=gp.exchanged_bytes["factor_input_received_per_rank"], iters=gp.cg_iters)
        ops.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,m", [(2, 2048 + 77, 70), (1, 1500, 40), (3, 1000, 33)])
def test_row_sharded_float32_exchange_on_the_gpu_matches_single_gpu_and_oracle(tmp_path, world, n, m):
    """SURVEY.md 8e with the exchange it sizes (round 4, nngp-src_amd/shard32.py): every rank keeps its float64 kernel rows, ONE
    all-gather carries the float32 factor input, the CG's matrix-vector product and the covariance's residual product are sharded by the
    rows a rank holds.  HIP kernels do the per-rank work (kernel rows, factor input, MFMA Cholesky, preconditioner, blocked solves,
    float64 GEMM), gloo the collectives between ranks sharing the test box's one GPU.  alpha, means and level-1 variances against the
    single-GPU model AND the float64 oracle; the bytes received for the factor input are half the float64 kernel's."""
    d = 24
    sys.path.insert(0, ROOT)
    from nngp_src_amd import synth
    from nngp_src_amd.model import GPModel
    mp.spawn(_worker_shard32, args=(world, _free_port(), n, m, d, str(tmp_path)), nprocs=world, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(m, d, seed=1)
    ref = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    m0, v0 = ref.predict(xt, cov="diag")
    a0 = ref.alpha().cpu().numpy().ravel()
    for p_ in (ROOT, os.path.join(ROOT, "oracle")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import nngp_oracle as o
    post = o.Posterior(x, y, o.make_arch(1), diag_reg=1e-3)
    m_or, c_or = post.predict(xt, "nngp", True)
    a_or = post._factor("nngp")[2].ravel()
    chunk = (n + world - 1) // world
    for r in range(world):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.linalg.norm(g["alpha"] - a0) <= 1e-8 * np.linalg.norm(a0)
        assert np.linalg.norm(g["alpha"] - a_or) <= 1e-7 * np.linalg.norm(a_or)
        assert np.linalg.norm(g["mean"] - np.ravel(m0)) <= 1e-8 * np.linalg.norm(m0)
        assert np.linalg.norm(g["m_model"] - np.ravel(m0)) <= 1e-8 * np.linalg.norm(m0)
        assert np.linalg.norm(g["mean"] - m_or.ravel()) <= 1e-6 * np.linalg.norm(m_or)   # north-star gate: 1e-4
        np.testing.assert_allclose(g["var"], v0, rtol=1e-4)
        np.testing.assert_allclose(g["var"], np.diag(c_or), rtol=3e-4)                    # gate: 1e-3
        if world > 1:
            ld = ((chunk * world + 127) // 128) * 128
            assert int(g["recv"]) == (world - 1) * chunk * ld * 4


def _worker2d(rank, world, port, pr, pc, n, d, nb, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nngp_src_amd import dist2d, synth
        x, y = synth.synthetic_queries(n, d, seed=0)
        xt, _ = synth.synthetic_queries(70, d, seed=1)
        grid = dist2d.Grid(pr, pc)
        gp = dist2d.Dist2DGP(dist2d.HipOps([1.0, 1.0], [0.0, 0.0]), grid, x, y, diag_reg=1e-3, nb=nb)
        gp.h3_min_tiles = 1  # every trailing update after block column 0 on the float16 pipe, also at this small size
        gp.fit()
        nbk = (n + nb - 1) // nb
        assert gp.a32.shape == (len(range(grid.pr, nbk, pr)) * nb, len(range(grid.pc, nbk, pc)) * nb)  # its tiles only
        assert gp.clamped == 0 and gp.relres < 1e-10, (gp.clamped, gp.relres, gp.cg_iters)
        mean, var = gp.predict(xt)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, alpha=gp.alpha.cpu().numpy(), iters=gp.cg_iters)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb", [(2, 2, 2100, 256), (2, 1, 1300, 512)])
def test_2d_block_cyclic_fit_on_the_gpu_matches_single_gpu_and_oracle(tmp_path, pr, pc, n, nb):
    """the 2-D block-cyclic distributed fit (nngp-src_amd/dist2d.py) with the HIP kernels doing the tile
    work -- kernel build of the rank's own tiles, leaf Cholesky + inverse, float32 and float64 MFMA GEMMs -- and gloo doing
    the collectives between ranks that share the test box's one GPU.  No rank holds the whole kernel or factor; alpha, means
    and level-1 variances must agree with the single-GPU model."""
    d, world = 24, pr * pc
    sys.path.insert(0, ROOT)
    from nngp_src_amd import synth
    from nngp_src_amd.model import GPModel
    mp.spawn(_worker2d, args=(world, _free_port(), pr, pc, n, d, nb, str(tmp_path)), nprocs=world, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(70, d, seed=1)
    ref = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    m0, v0 = ref.predict(xt, cov="diag")
    a0 = ref.alpha().cpu().numpy().ravel()
    # the checker: the float64 oracle on the same data (not only the single-GPU HIP model)
    for p_ in (ROOT, os.path.join(ROOT, "oracle")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import nngp_oracle as o
    post = o.Posterior(x, y, o.make_arch(1), diag_reg=1e-3)
    m_or, c_or = post.predict(xt, "nngp", True)
    a_or = post._factor("nngp")[2].ravel()
    for r in range(world):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.linalg.norm(g["alpha"] - a0) / np.linalg.norm(a0) < 1e-7
        assert np.linalg.norm(g["mean"] - m0.ravel()) / np.linalg.norm(m0) < 1e-8
        np.testing.assert_allclose(g["var"], v0, rtol=1e-4)
        assert np.linalg.norm(g["alpha"] - a_or) / np.linalg.norm(a_or) < 1e-6
        assert np.linalg.norm(g["mean"] - m_or.ravel()) / np.linalg.norm(m_or) < 1e-6  # north-star gate: 1e-4
        np.testing.assert_allclose(g["var"], np.diag(c_or), rtol=3e-4)                    # gate: 1e-3


def _bench_line(argv, env_extra, timeout=900):
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE line on stdout, whatever the ranks printed
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 2` with NO launcher around it and no WORLD_SIZE in the environment -- the way the driver ran the
    N = 1 record -- starts its ranks as a child torch.distributed.run, relays one parsable line and exits 0.  Two ranks share the
    test box's one GPU (gloo); the line must say two ranks took part and carry the alpha of the one-rank run."""
    import json
    argv = ["--config", "cfg1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    two = _bench_line(["--gpus", "2"] + argv, {"NNGP_DIST_BACKEND": "gloo"})
    one = _bench_line(["--gpus", "1"] + argv, {})
    assert two["n_gpus"] == 2 and two["shard"]["ranks_seen"] == 2 and one["n_gpus"] == 1
    assert two["config"]["N"] == one["config"]["N"] == 1000
    assert abs(two["fit_info"]["alpha_l2"] - one["fit_info"]["alpha_l2"]) <= 1e-9 * one["fit_info"]["alpha_l2"]
    assert two["fit_info"]["clamped_pivots"] == 0 and two["fit_info"]["rel_residual"] < 1e-10
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r3_bench_selflaunch_2rank_gloo.json"), "w") as f:
            json.dump({"two_ranks": two, "one_rank": one}, f, indent=1)
    # without a device for every rank and without an explicit backend the child falls back to the rehearsal backend and says so
    auto = _bench_line(["--gpus", "2", "--no-compare"] + argv, {})
    assert auto["shard"]["ranks_seen"] == 2 and ("gloo" in auto["shard"]["collective"] or torch.cuda.device_count() >= 2)
