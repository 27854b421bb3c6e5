"""world_size-2 (and 3) gloo rehearsal of the multi-GPU path on CPU: row partition + one all-gather.

The HIP kernel cannot run here, so the per-rank block builder is the oracle (tests may use it as a
stand-in); what is under test is the sharding and the collective in nngp-src_amd/distributed.py.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nngp_oracle as o
        from nngp_src_amd import distributed, synth

        arch = o.make_arch(1)
        x, _ = synth.synthetic_queries(n, d, seed=0)

        def oracle_kernel_fn(x1, x2=None, get="nngp", rows=None):
            r0, r1 = rows if rows is not None else (0, x1.shape[0])
            return o.kernel_fn(x1[r0:r1], x1 if x2 is None else x2, get, arch)

        K = distributed.sharded_kernel(oracle_kernel_fn, x, None, "nngp")
        full = o.kernel_fn(x, None, "nngp", arch)
        assert K.shape == (n, n)
        # blockwise BLAS calls may round differently from one full matmul: compare to 1e-13, gather itself is exact below
        np.testing.assert_allclose(K, full, rtol=1e-13, atol=1e-13 * np.abs(full).max())

        # in-place buffer all-gather with a padded leading dimension, as the GPU path does on the HBM buffer
        c = distributed.row_chunk(n, world)
        ld = n + 5
        buf = torch.full((world * c + 3, ld), float("nan"), dtype=torch.float64)
        r0, r1 = distributed.row_partition(n, world, rank)
        buf[r0:r1, :n] = torch.from_numpy(full[r0:r1])
        distributed.allgather_rows(buf, n)
        assert np.array_equal(buf[:n, :n].numpy(), full)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 70), (2, 64), (3, 128)])
def test_sharded_kernel_allgather_gloo(tmp_path, world, n):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, 6, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_bench_self_launcher_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE starts its ranks as a child torch.distributed.run (the parent
    never imports torch); on a host without a GPU the ranks stop with a message, nothing reaches stdout, and the parent returns the
    failure.  (The successful two-rank line is tests/test_gpu_distributed.py::test_bench_launches_its_own_ranks.)"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU host: covered by the -m gpu test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["NNGP_COLLECTIVE"] = "torch"  # no second attempt: there is no GPU either way
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg1", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no GPU visible" in r.stderr and "2-rank child run" in r.stderr
