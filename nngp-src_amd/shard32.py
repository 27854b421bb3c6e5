"""Row-sharded exact GP (the ``nt.batch(kernel_fn, device_count=G)`` slot of the reference, train.py:166-168) with the
exchange 8e sizes: what travels between GPUs is the FLOAT32 factor input, not the float64 kernel.

Round 3's multi-GPU fit all-gathered the float64 kernel (8 N^2 / G bytes per rank: 4.3 GB at N = 65536, G = 8 -- twice what SURVEY 8e
budgets, and by the repo's own cost model slower than rebuilding the kernel locally).  Here

* rank g builds its rows ``[g c, (g+1) c)`` of the float64 kernel (``nngp_model_build_rows``) and KEEPS them;
* it converts them to the float32 factor input (``nngp_model_factor_input_rows``) and ONE in-place all-gather of that buffer
  (4 N^2 / G bytes per rank: 2.15 GB at cfg4) makes the factorisation's input whole on every rank; the factorisation itself is the
  single-GPU one, replicated, or the 1-D block-cyclic one of ``distributed.distributed_factor``;
* alpha = (K + reg I)^-1 y by preconditioned CG driven from here: the preconditioner is the replicated float32 factor
  (``nngp_model_precond``), the matrix-vector product is SHARDED by the rows a rank holds (``nngp_model_matvec_rows``, 8 N^2 / G bytes
  of HBM per iteration and rank instead of 4 N^2) followed by one all-gather of an N-vector;
* the posterior of M test rows (NNGP): test rows are dealt to ranks; each rank solves its rows against the factor
  (``nngp_model_apply_factor``), the ranks all-gather those float32 rows Z (4 M N bytes in all), every rank multiplies ALL of them by
  its own kernel rows -- ``(Z A)[:, rows of g] = Z K[rows of g, :]^T``, K symmetric: the float64 residual product sharded by the rows a
  rank holds -- and one all-gather of the M x N/G blocks returns the products; the level-1 variance formula of the single-GPU path
  follows locally (``k^T A^-1 k = z.(k + r) + |L^-1 r|^2``, csrc/api_predict.hip).

The tile arithmetic sits behind a small ``ops`` interface (``HipRowOps``: the C ABI on the MI355X; the tests also run the distribution
logic with a NumPy stand-in on CPU ranks).  UNMEASURED ON HARDWARE: no multi-GPU box was available to rounds 1-4; rehearsed with gloo
(world 2 and 3 on CPU, 2 ranks sharing one GPU).  ``bench.py --mode shard32`` runs it.
"""
from __future__ import annotations

import numpy as np

from . import distributed as D


class HipRowOps:
    """The per-rank arithmetic of the row-sharded layout on one MI355X through the C ABI.  Raises without the library or a GPU."""

    def __init__(self, n: int, d: int, w_std, b_std, diag_reg: float = 1e-3, diag_reg_absolute_scale: bool = False, world: int = 1):
        import ctypes
        import torch
        from . import _lib
        from .model import GPModel
        self._lib, self._ct, self.torch = _lib, ctypes, torch
        self.lib = _lib.load()
        self.device = _lib.require_gpu()
        self.n, self.d = int(n), int(d)
        n_cap = D.row_chunk(n, world) * world
        self.model = GPModel(n_cap, d, w_std, b_std, get="nngp", diag_reg=diag_reg, diag_reg_absolute_scale=diag_reg_absolute_scale)
        self._zpad = None  # padded float64 copy of the current Z block (rows_times)
        self.arch = _lib.make_arch(w_std, b_std)
        self._x = None

    # -- data
    def to_device(self, a, dtype=None):
        t = a if isinstance(a, self.torch.Tensor) else self.torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def to_host(self, t):
        return t.detach().cpu().numpy()

    def set_train(self, x, y):
        self.model.set_train(x, y)
        self._x = self.to_device(np.asarray(x, dtype=np.float64))
        return self.model.info()["reg"]

    # -- fit
    def build_rows(self, r0, r1):
        self.model.build_rows(r0, r1)

    def factor_input_rows(self, r0, r1, shift_scale=1.0):
        L = self._lib
        L.check(self.lib.nngp_model_factor_input_rows(self.model.handle, int(r0), int(r1), float(shift_scale), L.stream_ptr()), self.lib)

    def factor_input_buffer(self):
        """([rows the allocation holds, ld] float32 torch view of the factor input / factor)."""
        L, ct = self._lib, self._ct
        a, ld, dinv = ct.c_void_p(), ct.c_int64(), ct.c_void_p()
        L.check(self.lib.nngp_model_factor_buffers(self.model.handle, ct.byref(a), ct.byref(ld), ct.byref(dinv)), self.lib)
        np_cap = (self.model.n_cap + 127) // 128 * 128
        rows = (np_cap * np_cap) // ld.value
        from .model import _wrap_device
        return _wrap_device(a.value, rows * ld.value, self.device, "<f4").view(rows, ld.value)

    def factor(self, group=None, comm=None, distributed_cholesky=False):
        """Every row of the factor input is in place.  Returns the clamped-pivot count (0: a sound float32 factor)."""
        L = self._lib
        L.check(self.lib.nngp_model_factor_input_complete(self.model.handle), self.lib)
        if distributed_cholesky and D.world_size() > 1:
            D.distributed_factor(self.model, group, comm=comm)
        else:
            self.model.factor()
        return int(self.model.info()["clamped_pivots"])

    def precond(self, r):
        L = self._lib
        z = self.torch.empty_like(r)
        L.check(self.lib.nngp_model_precond(self.model.handle, L.ptr(r), L.ptr(z), L.stream_ptr()), self.lib)
        return z

    def matvec_rows(self, p, r0, r1):
        L = self._lib
        q = self.torch.empty((r1 - r0,), dtype=self.torch.float64, device=self.device)
        if r1 > r0:
            L.check(self.lib.nngp_model_matvec_rows(self.model.handle, L.ptr(p), L.ptr(q), int(r0), int(r1), L.stream_ptr()), self.lib)
        return q

    def set_alpha(self, alpha, iters, relres):
        L = self._lib
        L.check(self.lib.nngp_model_set_alpha(self.model.handle, L.ptr(alpha), int(iters), float(relres), L.stream_ptr()), self.lib)

    # -- posterior
    def cross(self, xt):
        """float64 K(xt, X) [m, n]."""
        L, ct = self._lib, self._ct
        m = xt.shape[0]
        out = self.torch.empty((m, self.n), dtype=self.torch.float64, device=self.device)
        if m:
            L.check(self.lib.nngp_kernel_build(L.ptr(xt), m, L.ptr(self._x), self.n, self.d, ct.byref(self.arch), L.DTYPE_F64, L.ptr(out),
                                               None, self.n, 0, m, L.stream_ptr()), self.lib)
        return out

    def diag(self, xt):
        L, ct = self._lib, self._ct
        m = xt.shape[0]
        dn = self.torch.empty((m,), dtype=self.torch.float64, device=self.device)
        dt = self.torch.empty((m,), dtype=self.torch.float64, device=self.device)
        if m:
            L.check(self.lib.nngp_kernel_diag(L.ptr(xt), m, self.d, ct.byref(self.arch), L.ptr(dn), L.ptr(dt), L.stream_ptr()), self.lib)
        return dn

    def apply_factor(self, b32, both_halves):
        """b32 [rows, n] float32, returns b L^-T (or b (L L^T)^-1) as a new tensor."""
        out = b32.clone().contiguous()
        if out.shape[0]:
            self.model.apply_factor(out, both_halves=both_halves)
        return out

    def rows_times(self, z, r0, r1):
        """float64 [rows of z, r1 - r0] = z K[r0:r1, :]^T with this rank's kernel rows (float64 MFMA GEMM)."""
        L = self._lib
        kbuf, ld = self.model.kernel_buffer(all_rows=True)
        m = z.shape[0]
        nr = r1 - r0
        out = self.torch.zeros((m, max(nr, 0)), dtype=self.torch.float64, device=self.device)
        if m == 0 or nr <= 0:
            return out
        # the GEMM works on multiples of 128: pad the operands (zero rows / columns contribute nothing)
        mp, nrp, kp = -(-m // 128) * 128, -(-nr // 128) * 128, -(-self.n // 128) * 128
        # one padded copy of z, kept between calls; the kernel rows are read where they lie (their padding columns are zero; rows past
        # r1 inside the last 128-row tile only feed output columns that are cut off below) unless the tile would leave the buffer
        if self._zpad is None or self._zpad.shape != (mp, kp):
            self._zpad = self.torch.zeros((mp, kp), dtype=self.torch.float64, device=self.device)
        zp = self._zpad
        zp[:m, :self.n] = z
        if mp > m:
            zp[m:].zero_()
        if r0 + nrp <= kbuf.shape[0] and kp <= ld:
            kp_rows, kld = kbuf[r0:], ld
        else:
            kp_rows = self.torch.zeros((nrp, kp), dtype=self.torch.float64, device=self.device)
            kp_rows[:nr, :self.n] = kbuf[r0:r1, :self.n]
            kld = kp
        c = self.torch.zeros((mp, nrp), dtype=self.torch.float64, device=self.device)
        L.check(self.lib.nngp_gemm_nt_f64(L.ptr(c), nrp, L.ptr(c), nrp, L.ptr(zp), kp, L.ptr(kp_rows), kld, mp, nrp, kp, 1.0, 0.0,
                                          L.stream_ptr()), self.lib)
        out.copy_(c[:m, :nr])
        return out

    def close(self):
        self.model.close()


class RowShardedGP:
    """Exact NNGP posterior with the kernel sharded by rows (see the module text).  ``x``, ``y``: host arrays, the same on every rank."""

    def __init__(self, ops, x, y, group=None, comm=None, distributed_cholesky=False, max_iters=60, tol=1e-10):
        import torch
        self.torch, self.ops, self.group, self.comm = torch, ops, group, comm
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64).reshape(self.x.shape[0])
        self.n = self.x.shape[0]
        self.world, self.rank = D.world_size(), D.rank()
        self.r0, self.r1 = D.row_partition(self.n, self.world, self.rank)
        self.chunk = D.row_chunk(self.n, self.world)
        self.distributed_cholesky = distributed_cholesky
        self.max_iters, self.tol = max_iters, tol
        self.reg = None
        self.alpha = None
        self.cg_iters, self.relres, self.shift_scale = 0, 0.0, 1.0
        self.exchanged_bytes = {"factor_input_received_per_rank": 0, "cg_vectors_received_per_rank": 0}

    # -- collectives: nccl (= RCCL over xGMI) on device tensors; gloo (the rehearsals) stages device tensors through the host
    def _gather_blocks(self, mine):
        """mine [R, w] (the same shape on every rank) -> [world, R, w] on every rank."""
        t = self.torch
        if self.world == 1:
            return mine[None]
        dist = D._dist()
        mine = mine.contiguous()
        if mine.is_cuda and dist.get_backend(self.group) != "nccl":
            host = mine.cpu()
            parts = [t.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host, group=self.group)
            return t.stack(parts).to(mine.device)
        out = t.empty((self.world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)  # concatenated along dim 0
        dist.all_gather_into_tensor(out, mine, group=self.group)
        return out.view((self.world,) + tuple(mine.shape))

    def _allgather_vec(self, local, total):
        """local: this rank's [r1 - r0] slice -> the whole [total] vector on every rank."""
        t = self.torch
        if self.world == 1:
            return local
        mine = t.zeros((self.chunk, 1), dtype=local.dtype, device=local.device)
        mine[:local.shape[0], 0] = local
        self.exchanged_bytes["cg_vectors_received_per_rank"] += (self.world - 1) * self.chunk * local.element_size()
        return self._gather_blocks(mine).reshape(-1)[:total]

    def _allgather_rows(self, local, rows_per_rank):
        """local [<= rows_per_rank, w] -> [world, rows_per_rank, w] on every rank (short blocks are zero padded)."""
        t = self.torch
        mine = t.zeros((rows_per_rank, local.shape[1]), dtype=local.dtype, device=local.device)
        mine[:local.shape[0]] = local
        return self._gather_blocks(mine)

    # -- fit
    def fit(self):
        ops, t = self.ops, self.torch
        self.reg = float(ops.set_train(self.x, self.y))
        ops.build_rows(self.r0, self.r1)
        shift, factored_shift, clamped = 1.0, 1.0, 0
        for attempt in range(5):
            factored_shift = shift
            ops.factor_input_rows(self.r0, self.r1, shift)
            if self.world > 1:
                buf = ops.factor_input_buffer()
                D.allgather_rows(buf, self.n, self.group, self.comm)
                self.exchanged_bytes["factor_input_received_per_rank"] += (self.world - 1) * self.chunk * buf.shape[1] * buf.element_size()
            clamped = ops.factor(self.group, self.comm, self.distributed_cholesky)
            # float32 breakdown (cond(K + reg I) eps32 >> 1): every rank saw the same factor input, hence the same count -- all of them
            # redo the exchange with a 16x larger shift (the factor is only the CG's preconditioner), as nngp_model_factor does alone
            if clamped == 0:
                break
            shift *= 16.0
        if clamped != 0:  # (nngp_model_factor alone gives up the same way: a preconditioner with clamped pivots is not trusted)
            raise RuntimeError("shard32.fit: the float32 factor still has %d clamped pivots with the regulariser shift raised %g-fold"
                               % (clamped, factored_shift))
        self.shift_scale = factored_shift  # the shift that WAS factored (it scales the CG's iteration limit)
        self.alpha = self._pcg(ops.to_device(self.y))
        ops.set_alpha(self.alpha, self.cg_iters, self.relres)
        return self

    def _matvec(self, p):
        q_local = self.ops.matvec_rows(p, self.r0, self.r1)
        q_local = q_local + self.reg * p[self.r0:self.r1]
        return self._allgather_vec(q_local, self.n)

    def _pcg(self, b):
        """Preconditioned CG in float64; every rank runs the same scalar recurrences on the same gathered vectors."""
        ops = self.ops
        x = self.torch.zeros_like(b)
        r = b.clone()
        bnorm = float(self.torch.linalg.vector_norm(b))
        if bnorm == 0.0:
            return x
        z = ops.precond(r)
        p = z.clone()
        rz = float(self.torch.dot(r, z))
        limit = int(self.max_iters * max(1.0, np.sqrt(self.shift_scale)))
        for it in range(limit):
            q = self._matvec(p)
            pq = float(self.torch.dot(p, q))
            a = rz / pq if pq != 0.0 else 0.0
            x = x + a * p
            r = r - a * q
            self.cg_iters = it + 1
            self.relres = float(self.torch.linalg.vector_norm(r)) / bnorm
            if not np.isfinite(self.relres):
                raise RuntimeError("row-sharded CG: residual became NaN at iteration %d" % (it + 1))
            if self.relres <= self.tol:
                break
            z = ops.precond(r)
            rz_new = float(self.torch.dot(r, z))
            p = z + (rz_new / rz) * p
            rz = rz_new
        return x

    # -- posterior
    def predict(self, xt, cov=True):
        """(mean [M], variance [M] or None) on every rank.  ``xt``: the same host array on every rank; its rows are dealt to ranks."""
        ops, t = self.ops, self.torch
        xt = np.ascontiguousarray(xt, dtype=np.float64)
        m = xt.shape[0]
        mc = D.row_chunk(m, self.world)
        m0, m1 = D.row_partition(m, self.world, self.rank)
        xh = ops.to_device(xt[m0:m1])
        ktd = ops.cross(xh)                                  # [m_h, n] float64
        mean_h = ktd @ self.alpha
        var_h = None
        if cov:
            z_h = ops.apply_factor(ktd.to(t.float32), True)   # rows K_td (L L^T)^-1, float32
            z_all = self._allgather_rows(z_h, mc).reshape(self.world * mc, self.n)   # every rank's rows (padding rows are zero)
            part = ops.rows_times(z_all.to(t.float64), self.r0, self.r1)            # [world mc, r1 - r0] = Z K[rows of this rank, :]^T
            full = self._allgather_rows(part.t().contiguous(), self.chunk)           # [world, chunk, world mc]: (Z A)^T by column blocks
            za = full.reshape(self.world * self.chunk, self.world * mc)[:self.n, self.rank * mc:self.rank * mc + (m1 - m0)].t()
            zh64 = z_h.to(t.float64)
            r = ktd - za - self.reg * zh64                    # r0 = k - (K + reg I) z0, float64
            v = ops.apply_factor(r.to(t.float32), False).to(t.float64)   # L^-1 r0
            var_h = ops.diag(xh) - (zh64 * (ktd + r)).sum(dim=1) - (v * v).sum(dim=1)
        mean = self._allgather_rows(mean_h[:, None], mc).reshape(-1)[:self.world * mc]
        mean = t.cat([mean[g * mc:g * mc + (D.row_partition(m, self.world, g)[1] - D.row_partition(m, self.world, g)[0])] for g in range(self.world)])
        if not cov:
            return ops.to_host(mean), None
        var = self._allgather_rows(var_h[:, None], mc).reshape(-1)
        var = t.cat([var[g * mc:g * mc + (D.row_partition(m, self.world, g)[1] - D.row_partition(m, self.world, g)[0])] for g in range(self.world)])
        return ops.to_host(mean), ops.to_host(var)
