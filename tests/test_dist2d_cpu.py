"""gloo rehearsal on CPU of the 2-D block-cyclic distributed GP (nngp-src_amd/dist2d.py).

The HIP kernels cannot run here, so the tile arithmetic is a torch-CPU / oracle stand-in with the interface of
``dist2d.HipOps`` (tests may use the oracle as a stand-in); what is under test is everything else: tile ownership, the
broadcast / all-gather / reduction pattern of the factorisation, the fan-in triangular solves on replicated right-hand sides,
the distributed float64 matrix-vector product, the CG for alpha and the level-1 variance -- on 2 x 2, 2 x 1, 1 x 2 and 3 x 2
process grids, with N not a multiple of the tile size.
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


class CpuOps:
    """Stand-in for dist2d.HipOps: same methods, torch CPU tensors, kernel from the float64 oracle."""

    def __init__(self, arch, get="nngp"):
        import nngp_oracle as o
        self.o, self.arch, self.get = o, arch, get

    def to_device(self, a, dtype=None):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype=dtype or t.dtype).contiguous()

    def kernel(self, x1, x2):
        if x1.shape[0] == 0 or x2.shape[0] == 0:
            return torch.empty((x1.shape[0], x2.shape[0]), dtype=torch.float64)
        return torch.from_numpy(self.o.kernel_fn(x1.numpy(), x2.numpy(), self.get, self.arch))

    def kernel_diag(self, x):
        q = (x * x).sum(dim=1).numpy() / x.shape[1]
        kd, td = self.o.diag_kernel(q, self.arch)
        return torch.from_numpy(kd), torch.from_numpy(kd if self.get == "nngp" else td)

    def potrf_inverse(self, tile):
        l = torch.linalg.cholesky(torch.tril(tile.double()) + torch.tril(tile.double(), -1).T)
        tile.copy_((torch.tril(l) + torch.triu(tile.double(), 1)).float())  # lower part replaced, upper entries kept (like the leaf)
        minv = torch.linalg.inv(l)
        return minv.float().contiguous(), minv.T.float().contiguous(), 0

    def gemm_nt(self, c, a, b, alpha, beta):
        if a.shape[0] and b.shape[0] and a.shape[1]:
            c.copy_(beta * c + alpha * (a @ b.T))
        return c

    gemm_nt64 = gemm_nt


def _worker(rank, world, port, pr, pc, n, d, nb, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nngp_oracle as o
        from nngp_src_amd import dist2d, synth
        arch = o.make_arch(1)
        x, y = synth.synthetic_queries(n, d, seed=0)
        xt, _ = synth.synthetic_queries(37, d, seed=1)
        grid = dist2d.Grid(pr, pc)
        gp = dist2d.Dist2DGP(CpuOps(arch), grid, x, y, diag_reg=1e-3, nb=nb).fit()
        # no rank holds the whole matrix: local storage is its share of the tiles only
        nbk = (n + nb - 1) // nb
        assert gp.a32.shape == (len(range(grid.pr, nbk, pr)) * nb, len(range(grid.pc, nbk, pc)) * nb)
        mean, var = gp.predict(xt)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, alpha=gp.alpha.numpy(), iters=gp.cg_iters, relres=gp.relres)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pr,pc,n,nb", [(2, 2, 700, 128), (2, 1, 515, 128), (1, 2, 515, 128), (3, 2, 1000, 128), (2, 2, 300, 256)])
def test_2d_block_cyclic_fit_matches_the_oracle(tmp_path, pr, pc, n, nb):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nngp_oracle as o
    from nngp_src_amd import synth
    d, world = 12, pr * pc
    mp.spawn(_worker, args=(world, _free_port(), pr, pc, n, d, nb, str(tmp_path)), nprocs=world, join=True)
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(37, d, seed=1)
    post = o.Posterior(x, y, o.make_arch(1), diag_reg=1e-3)
    mean_ref, cov_ref = post.predict(xt, "nngp", True)
    alpha_ref = post._factor("nngp")[2].ravel()
    for r in range(world):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert g["relres"] < 1e-10 and g["iters"] <= 12, (g["relres"], g["iters"])
        assert np.linalg.norm(g["alpha"] - alpha_ref) / np.linalg.norm(alpha_ref) < 1e-7
        assert np.linalg.norm(g["mean"] - mean_ref.ravel()) / np.linalg.norm(mean_ref) < 1e-8
        np.testing.assert_allclose(g["var"], np.diag(cov_ref), rtol=1e-4)


def test_single_process_grid_is_the_same_code():
    """world = 1 (no process group): the same class without any collective."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nngp_oracle as o
    from nngp_src_amd import dist2d, synth
    n, d = 400, 10
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(9, d, seed=1)
    gp = dist2d.Dist2DGP(CpuOps(o.make_arch(2)), dist2d.Grid(1, 1), x, y, diag_reg=1e-3, nb=128).fit()
    mean, var = gp.predict(xt)
    m_ref, c_ref = o.Posterior(x, y, o.make_arch(2), diag_reg=1e-3).predict(xt, "nngp", True)
    assert np.linalg.norm(mean - m_ref.ravel()) / np.linalg.norm(m_ref) < 1e-8
    np.testing.assert_allclose(var, np.diag(c_ref), rtol=1e-4)
