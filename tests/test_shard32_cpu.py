"""The row-sharded layout (nngp-src_amd/shard32.py: float32 factor input exchanged, float64 kernel rows kept local, CG matrix-vector
product and the covariance's residual product sharded by those rows) on gloo ranks WITHOUT a GPU: the per-rank arithmetic is a NumPy
stand-in built on the oracle (tests may use it as the checker and as a stand-in); what is under test is the distribution logic --
row partitions with short and empty tail blocks, the in-place float32 all-gather, the CG's gathered vectors, the two gathers of the
posterior -- against the oracle's posterior on the whole problem."""
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


class NumpyRowOps:
    """CPU stand-in of shard32.HipRowOps: float64 kernel rows from the oracle, a float32 Cholesky of the gathered factor input."""

    def __init__(self, oracle, arch, n, world, diag_reg=1e-3):
        self.o, self.arch, self.n, self.world, self.diag_reg = oracle, arch, n, world, diag_reg
        self.chunk = (n + world - 1) // world
        self.a32 = torch.full((world * self.chunk, n + 3), float("nan"), dtype=torch.float32)  # padded leading dimension, like the HBM buffer
        self.gathers = 0

    def to_device(self, a, dtype=None):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype=dtype or t.dtype).contiguous()

    def to_host(self, t):
        return t.numpy()

    def set_train(self, x, y):
        self.x = x
        q = np.sum(x * x, axis=1) / x.shape[1]
        self.reg = self.diag_reg * float(np.mean(self.o.diag_kernel(q, self.arch)[0]))
        return self.reg

    def build_rows(self, r0, r1):
        self.r0, self.r1 = r0, r1
        self.krows = self.o.kernel_fn(self.x[r0:r1], self.x, "nngp", self.arch) if r1 > r0 else np.zeros((0, self.n))

    def factor_input_rows(self, r0, r1, shift_scale=1.0):
        blk = self.krows.copy()
        blk[np.arange(r1 - r0), np.arange(r0, r1)] += self.reg * shift_scale
        self.a32[r0:r1, :self.n] = torch.from_numpy(blk.astype(np.float32))

    def factor_input_buffer(self):
        return self.a32

    def factor(self, group=None, comm=None, distributed_cholesky=False):
        a = self.a32[:self.n, :self.n].numpy()
        assert np.isfinite(a).all(), "rows of the factor input are missing after the all-gather"
        import scipy.linalg as sla
        self.L = sla.cholesky(np.tril(a) + np.tril(a, -1).T, lower=True).astype(np.float32)
        return 0

    def _solve(self, b, trans):
        import scipy.linalg as sla
        return sla.solve_triangular(self.L, b, lower=True, trans=trans, check_finite=False)

    def precond(self, r):
        v = self._solve(r.numpy().astype(np.float32), 0)
        return torch.from_numpy(self._solve(v, 1).astype(np.float64))

    def matvec_rows(self, p, r0, r1):
        return torch.from_numpy(self.krows @ p.numpy())

    def set_alpha(self, alpha, iters, relres):
        self.alpha_set = (alpha.clone(), iters, relres)

    def cross(self, xt):
        return torch.from_numpy(self.o.kernel_fn(xt.numpy(), self.x, "nngp", self.arch)) if xt.shape[0] else torch.zeros((0, self.n), dtype=torch.float64)

    def diag(self, xt):
        x = xt.numpy()
        return torch.from_numpy(self.o.diag_kernel(np.sum(x * x, axis=1) / x.shape[1], self.arch)[0]) if x.shape[0] else torch.zeros((0,), dtype=torch.float64)

    def apply_factor(self, b32, both_halves):
        if b32.shape[0] == 0:
            return b32.clone()
        v = self._solve(b32.numpy().T.astype(np.float32), 0)       # L^-1 b^T
        if both_halves:
            v = self._solve(v, 1)
        return torch.from_numpy(np.ascontiguousarray(v.T.astype(np.float32)))

    def rows_times(self, z, r0, r1):
        return torch.from_numpy(z.numpy() @ self.krows.T)


def _worker(rank, world, port, n, m, d, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nngp_oracle as o
        from nngp_src_amd import shard32, synth
        arch = o.make_arch(1)
        x, y = synth.synthetic_queries(n, d, seed=0)
        xt, _ = synth.synthetic_queries(m, d, seed=1)
        x, xt = x / 1000.0, xt / 1000.0   # well inside float32's reach for the stand-in's plain float32 Cholesky
        ops = NumpyRowOps(o, arch, n, world)
        gp = shard32.RowShardedGP(ops, x, y).fit()
        assert gp.relres <= 1e-10 and gp.cg_iters <= 12, (gp.relres, gp.cg_iters)
        mean, var = gp.predict(xt, cov=True)
        mean_only, none = gp.predict(xt, cov=False)
        assert none is None and np.array_equal(mean, mean_only)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, alpha=gp.alpha.numpy(),
                 recv=gp.exchanged_bytes["factor_input_received_per_rank"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,m", [(2, 301, 37), (3, 256, 50), (3, 130, 2)])
def test_row_sharded_layout_on_gloo_ranks_matches_the_oracle(tmp_path, world, n, m):
    d = 8
    mp.spawn(_worker, args=(world, _free_port(), n, m, d, str(tmp_path)), nprocs=world, join=True)
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nngp_oracle as o
    from nngp_src_amd import synth
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(m, d, seed=1)
    x, xt = x / 1000.0, xt / 1000.0
    post = o.Posterior(x, y, o.make_arch(1), diag_reg=1e-3)
    m_ref, c_ref = post.predict(xt, "nngp", True)
    a_ref = post._factor("nngp")[2].ravel()
    chunk = (n + world - 1) // world
    for r in range(world):
        g = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.linalg.norm(g["alpha"] - a_ref) <= 1e-8 * np.linalg.norm(a_ref)
        assert np.linalg.norm(g["mean"] - m_ref.ravel()) <= 1e-8 * np.linalg.norm(m_ref)
        np.testing.assert_allclose(g["var"], np.diag(c_ref), rtol=1e-4)   # level-1 variance on a float32 factor: gate 1e-3
        # the exchange is the FLOAT32 factor input: (world - 1) chunks of 4-byte rows per fit and rank -- half the float64 kernel
        assert int(g["recv"]) == (world - 1) * chunk * (n + 3) * 4
