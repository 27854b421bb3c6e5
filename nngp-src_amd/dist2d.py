"""2-D block-cyclic distributed GP fit (no counterpart in the reference, which is single-process).

P = Pr x Pc processes, one per GPU.  The N x N kernel is cut into nb x nb tiles; tile (I, J) lives on process
(I mod Pr, J mod Pc) and NOWHERE else: no rank ever holds the whole kernel or the whole factor (memory per rank
12 N^2 / P bytes: the float64 kernel tiles it owns plus their float32 copy that becomes the factor).

* build      every rank builds exactly its own tiles with ONE rectangular launch of the kernel-build kernel on gathered rows
             of X (X itself is replicated: N d doubles) -- no N^2 collective at all;
* factor     right-looking over block columns k: the owner of (k, k) factors it (leaf Cholesky + explicit inverse of the
             nb-block) and broadcasts the inverse down its process COLUMN; that column's ranks solve their tiles of block column
             k (one GEMM with the inverse); each solved tile travels along its process ROW (one broadcast per process row) and
             the tiles a process column needs as the transposed operand are exchanged inside that column (one all-gather of
             exactly those tiles); every rank then updates its own trailing tiles with ONE GEMM;
* solves     X = B L^-T and Y = B L^-1 for a replicated block of right-hand-side rows, left-looking ("fan-in"): for block J the
             ranks of process row (column) J mod Pr (Pc) sum their tiles' contributions, one reduction inside that group brings
             them to the owner of (J, J), which applies the inverted diagonal block and broadcasts the result;
* alpha      preconditioned CG in float64: the matrix-vector product uses the distributed float64 tiles (one all-reduce of an
             N-vector per iteration), the preconditioner is the pair of distributed float32 solves;
* predict    means from alpha; variances by the level-1 formula of the single-GPU path on the same
             primitives: z0 = M^-1 k, r0 = k - A z0, var = K_tt - z0.(k + r0) - |L^-1 r0|^2.

All heavy arithmetic goes through an ``ops`` object: ``HipOps`` (below) drives the hand-written HIP kernels through the C ABI
(nngp_kernel_build, nngp_potrf_f32, nngp_trsm_rlt_f32, nngp_gemm_nt_f32, nngp_gemm_nt_f64); the CPU tests pass a stand-in with
the same methods so that the distribution logic -- ownership, collectives, fan-in solves -- is exercised by gloo on CPU.
Collectives: torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in tests and one-GPU rehearsals, where device
buffers are staged through the host).  UNMEASURED ON HARDWARE: no multi-GPU node was available; rehearsed with 2 x 2, 2 x 1,
1 x 2 and 3 x 2 grids (tests/test_dist2d_cpu.py, tests/test_gpu_distributed.py).
"""
from __future__ import annotations

import math

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


class Grid:
    """Pr x Pc process grid over the default group; rank = pr * Pc + pc."""

    def __init__(self, pr: int, pc: int):
        dist = _dist()
        self.Pr, self.Pc = int(pr), int(pc)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if self.world != self.Pr * self.Pc:
            raise ValueError("process grid %d x %d needs %d ranks, the group has %d" % (pr, pc, pr * pc, self.world))
        self.pr, self.pc = self.rank // self.Pc, self.rank % self.Pc
        # every rank creates every subgroup, in the same order (torch.distributed requirement)
        self.row_groups = [dist.new_group([r * self.Pc + c for c in range(self.Pc)]) for r in range(self.Pr)] if self.world > 1 else [None]
        self.col_groups = [dist.new_group([r * self.Pc + c for r in range(self.Pr)]) for c in range(self.Pc)] if self.world > 1 else [None]

    def owner(self, bi: int, bj: int) -> int:
        return (bi % self.Pr) * self.Pc + (bj % self.Pc)

    @property
    def row_group(self):
        return self.row_groups[self.pr]

    @property
    def col_group(self):
        return self.col_groups[self.pc]


class _Comm:
    """Collectives on device or host tensors; with gloo and device tensors the payload is staged through the host."""

    def __init__(self):
        self.dist = _dist()
        self.on = self.dist.is_initialized() and self.dist.get_world_size() > 1

    def _staged(self, t, group):
        return t.is_cuda and self.dist.get_backend(group) != "nccl"

    def bcast(self, t, src, group=None):
        if not self.on:
            return t
        if self._staged(t, group):
            h = t.cpu()
            self.dist.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            self.dist.broadcast(t, src=src, group=group)
        return t

    def reduce_sum(self, t, dst, group=None):
        """Sum over the group; the result is guaranteed on `dst` only."""
        if not self.on:
            return t
        if self._staged(t, group):
            h = t.cpu()
            self.dist.reduce(h, dst=dst, op=self.dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            self.dist.reduce(t, dst=dst, op=self.dist.ReduceOp.SUM, group=group)
        return t

    def allreduce_sum(self, t, group=None):
        if not self.on:
            return t
        if self._staged(t, group):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=group)
        return t

    def allgather(self, t, n, group=None):
        """[n] tensors shaped like t, one per rank of the group (group order)."""
        import torch
        if not self.on:
            return [t]
        if self._staged(t, group):
            h = t.cpu()
            parts = [torch.empty_like(h) for _ in range(n)]
            self.dist.all_gather(parts, h, group=group)
            return [p.to(t.device) for p in parts]
        parts = [torch.empty_like(t) for _ in range(n)]
        self.dist.all_gather(parts, t.contiguous(), group=group)
        return parts


class HipOps:
    """The tile arithmetic on the MI355X, through the C ABI (include/nngp_hip.h).  Raises without the library or a GPU."""

    def __init__(self, w_std, b_std, get: str = "nngp"):
        import ctypes
        import torch
        from . import _lib
        self._lib, self._ct, self.torch = _lib, ctypes, torch
        self.lib = _lib.load()
        self.device = _lib.require_gpu()
        self.arch = _lib.make_arch(w_std, b_std)
        self.get = get

    def to_device(self, a, dtype=None):
        t = a if isinstance(a, self.torch.Tensor) else self.torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=dtype, device=self.device)

    def kernel(self, x1, x2):
        """float64 kernel block K(x1, x2) of `get` ([n1, n2] device tensor; x1, x2 device float64)."""
        L, ct = self._lib, self._ct
        n1, d = x1.shape
        n2 = x2.shape[0]
        out = self.torch.empty((n1, n2), dtype=self.torch.float64, device=self.device)
        if n1 == 0 or n2 == 0:
            return out
        nn, nt = (out, None) if self.get == "nngp" else (None, out)
        L.check(self.lib.nngp_kernel_build(L.ptr(x1), n1, L.ptr(x2), n2, d, ct.byref(self.arch), L.DTYPE_F64, L.ptr(nn), L.ptr(nt), n2,
                                           0, n1, L.stream_ptr()), self.lib)
        return out

    def kernel_nngp(self, x1, x2):
        get, self.get = self.get, "nngp"
        try:
            return self.kernel(x1, x2)
        finally:
            self.get = get

    def kernel_diag(self, x):
        """(diag of the nngp kernel, diag of the `get` kernel) at the rows of x."""
        L, ct = self._lib, self._ct
        n, d = x.shape
        dn = self.torch.empty((n,), dtype=self.torch.float64, device=self.device)
        dt = self.torch.empty((n,), dtype=self.torch.float64, device=self.device)
        L.check(self.lib.nngp_kernel_diag(L.ptr(x), n, d, ct.byref(self.arch), L.ptr(dn), L.ptr(dt), L.stream_ptr()), self.lib)
        return dn, (dn if self.get == "nngp" else dt)

    def potrf_inverse(self, tile):
        """In place: lower Cholesky factor of the float32 nb x nb tile.  Returns (Minv = L^-1, T = L^-T, clamped pivots)."""
        L = self._lib
        nb = tile.shape[0]
        dinv = self.torch.empty((nb // 128, 128, 128), dtype=self.torch.float32, device=self.device)
        clamped = self.torch.zeros(1, dtype=self.torch.int32, device=self.device)
        L.check(self.lib.nngp_potrf_f32(L.ptr(tile), nb, tile.stride(0), L.ptr(dinv), L.ptr(clamped), L.stream_ptr()), self.lib)
        t = self.torch.eye(nb, dtype=self.torch.float32, device=self.device)
        L.check(self.lib.nngp_trsm_rlt_f32(L.ptr(t), nb, nb, L.ptr(tile), tile.stride(0), L.ptr(dinv), nb, L.stream_ptr()), self.lib)  # I L^-T
        return t.t().contiguous(), t, int(clamped.item())

    def gemm_nt(self, c, a, b, alpha, beta):
        """c = beta c + alpha a b^T (float32 MFMA; all dimensions multiples of 128; c must not alias a or b)."""
        L = self._lib
        m, k = a.shape
        n = b.shape[0]
        if m == 0 or n == 0 or k == 0:
            return c
        L.check(self.lib.nngp_gemm_nt_f32(L.ptr(c), c.stride(0), L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), m, n, k, float(alpha),
                                          float(beta), 0, L.stream_ptr()), self.lib)
        return c

    def gemm_nt_h3(self, c, a, b, alpha, beta, scale):
        """The same product on the float16 matrix pipe with float32-grade operands (nngp_gemm_nt_h3: hi + lo float16 planes,
        three products per term; `scale` a power of two with max |a|, |b| * scale < 2^15): the trailing updates of the
        distributed factorisation, 3/16 of the float32-MFMA time per term."""
        L = self._lib
        m, k = a.shape
        n = b.shape[0]
        if m == 0 or n == 0 or k == 0:
            return c
        L.check(self.lib.nngp_gemm_nt_h3(L.ptr(c), c.stride(0), L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), m, n, k, float(alpha),
                                         float(beta), float(scale), 0, L.stream_ptr()), self.lib)
        return c

    def gemm_nt64(self, c, a, b, alpha, beta):
        """The same in float64 (float64 MFMA)."""
        L = self._lib
        m, k = a.shape
        n = b.shape[0]
        if m == 0 or n == 0 or k == 0:
            return c
        L.check(self.lib.nngp_gemm_nt_f64(L.ptr(c), c.stride(0), L.ptr(c), c.stride(0), L.ptr(a), a.stride(0), L.ptr(b), b.stride(0),
                                          m, n, k, float(alpha), float(beta), L.stream_ptr()), self.lib)
        return c


class Dist2DGP:
    """Exact GP posterior (gradient_descent_mse_ensemble of the reference, train.py:171-172) with the kernel and its factor
    distributed 2-D block-cyclically.  ``x``, ``y``: host arrays, the same on every rank."""

    RHS = 128  # rows of a right-hand-side block (the GEMM kernels work on multiples of 128)

    def __init__(self, ops, grid: Grid, x, y, diag_reg: float = 1e-3, diag_reg_absolute_scale: bool = False, nb: int = 1024):
        import torch
        self.torch, self.ops, self.g, self.comm = torch, ops, grid, _Comm()
        if nb % 128 != 0 or nb <= 0:
            raise ValueError("nb must be a positive multiple of 128")
        self.n, self.d = int(np.shape(x)[0]), int(np.shape(x)[1])
        self.nb = int(nb)
        self.NB = (self.n + nb - 1) // nb
        self.Np = self.NB * nb
        self.x = ops.to_device(np.asarray(x, dtype=np.float64))
        self.y = ops.to_device(np.asarray(y, dtype=np.float64).reshape(self.n))
        self.rows = list(range(grid.pr, self.NB, grid.Pr))  # global block rows / columns of the local tiles, ascending
        self.cols = list(range(grid.pc, self.NB, grid.Pc))
        self.ridx = self._elem_index(self.rows)
        self.cidx = self._elem_index(self.cols)
        dn, dg = ops.kernel_diag(self.x)
        self.trace_mean = float(dg.mean().item())
        self.reg = float(diag_reg) if diag_reg_absolute_scale else float(diag_reg) * self.trace_mean
        # float16-pipe trailing updates: |L_ij| <= sqrt(max_i A_ii), scaled below 2^15 by a power of two (as SplitWork::scale)
        dmax = float(dg.max().item()) + self.reg + self.trace_mean
        self.h3_scale = 2.0 ** (14 - int(np.ceil(np.log2(max(np.sqrt(dmax), 1e-30)))))
        self.h3_min_tiles = 96
        self.minv, self.tinv = {}, {}  # L_JJ^-1 and L_JJ^-T of the diagonal tiles this rank owns
        self.clamped = 0
        self.alpha = None
        self.cg_iters, self.relres = 0, 0.0

    # ---- index helpers ----
    def _elem_index(self, blocks):
        idx = np.concatenate([np.arange(b * self.nb, (b + 1) * self.nb) for b in blocks]) if blocks else np.zeros((0,), dtype=np.int64)
        return self.ops.to_device(idx.astype(np.int64))

    def _local_pos(self, blocks, b):
        """Number of local blocks with global index < b (= local block position of b if it is local)."""
        return sum(1 for v in blocks if v < b)

    def _gather_rows(self, idx):
        """Rows of X at element indices idx; padding rows (index >= n) are zero vectors."""
        t = self.torch
        valid = idx < self.n
        out = t.zeros((idx.shape[0], self.d), dtype=t.float64, device=self.x.device)
        out[valid] = self.x[idx[valid]]
        return out, valid

    # ---- build: this rank's tiles only ----
    def build(self):
        t, nb = self.torch, self.nb
        xr, vr = self._gather_rows(self.ridx)
        xc, vc = self._gather_rows(self.cidx)
        # stored transposed (columns of the process grid along the rows of the local array): (P K)[:, J] needs K[:, J]^T = K[J, :]
        kt = self.ops.kernel(xc, xr)  # [lc, lr] = K[cidx, ridx]
        kt[~vc, :] = 0.0
        kt[:, ~vr] = 0.0
        self.k64t = kt
        a32 = kt.t().contiguous().to(t.float32)  # [lr, lc] = K[ridx, cidx]: becomes the factor (lower tiles)
        for i, bi in enumerate(self.rows):  # regulariser (and a decoupled diagonal in the padding) on this rank's diagonal tiles
            if bi % self.g.Pc == self.g.pc:
                j = self._local_pos(self.cols, bi)
                e = t.arange(nb, device=a32.device)
                gidx = bi * nb + e
                add = t.where(gidx < self.n, t.full((nb,), self.reg, dtype=t.float32, device=a32.device),
                              t.full((nb,), self.reg + self.trace_mean, dtype=t.float32, device=a32.device))
                a32[i * nb + e, j * nb + e] += add
        self.a32 = a32
        return self

    # ---- factor ----
    def factor(self):
        t, g, nb, ops, comm = self.torch, self.g, self.nb, self.ops, self.comm
        lr, lc = self.a32.shape
        for k in range(self.NB):
            prk, pck = k % g.Pr, k % g.Pc
            minv = t.empty((nb, nb), dtype=t.float32, device=self.a32.device)
            ik, jk = self._local_pos(self.rows, k), self._local_pos(self.cols, k)
            if g.pr == prk and g.pc == pck:  # diagonal tile: factor + invert
                tile = self.a32[ik * nb:(ik + 1) * nb, jk * nb:(jk + 1) * nb]
                mi, ti, cl = ops.potrf_inverse(tile)
                self.minv[k], self.tinv[k] = mi, ti
                self.clamped += cl
                minv.copy_(mi)
            i0 = self._local_pos(self.rows, k + 1)  # first local block row with I > k
            j0 = self._local_pos(self.cols, k + 1)
            mr, mc = lr - i0 * nb, lc - j0 * nb       # local trailing rows / columns
            prow = t.zeros((mr, nb), dtype=t.float32, device=self.a32.device)
            if g.pc == pck:
                comm.bcast(minv, src=g.owner(k, k), group=g.col_group)  # the inverse goes down process column pck
                if mr > 0:  # solve this rank's tiles of block column k: X = A L_kk^-T = A Minv^T
                    panel = self.a32[i0 * nb:, jk * nb:(jk + 1) * nb]  # a row-strided view: the GEMM takes leading dimensions
                    ops.gemm_nt(prow, panel, minv, 1.0, 0.0)
                    panel.copy_(prow)
            if mr > 0:
                comm.bcast(prow, src=g.pr * g.Pc + pck, group=g.row_group)  # solved tiles travel along their process row
            # transposed operand: tiles L_Jk with J = pc (mod Pc), J > k, collected inside this process column
            pcol = self._column_panel(k, prow, i0, j0, mc)
            if mr > 0 and mc > 0:  # ONE update of the local trailing tiles (upper ones included: never read)
                # on the float16 pipe from the second block column on, when the local tiles fill the GPU (>= 96 tiles of 256 x 256,
                # as in the single-GPU solves); block column 0 -- the kernel's dominant, one-signed columns -- stays float32
                # (accumulator truncation of the float16 pipe on same-sign sums: potrf.hip)
                h3 = getattr(ops, "gemm_nt_h3", None)
                # (h3_scale assumes |L_ij| <= sqrt(max A_ii); after a clamped pivot entries can exceed that and the float16 planes would
                # overflow to inf: from then on this rank's updates stay float32, so that the factor is finite and `clamped` can report it)
                if (h3 is not None and k > 0 and self.clamped == 0 and nb % 32 == 0
                        and ((mr + 255) // 256) * ((mc + 255) // 256) >= self.h3_min_tiles):
                    h3(self.a32[i0 * nb:, j0 * nb:], prow, pcol, -1.0, 1.0, self.h3_scale)
                else:
                    ops.gemm_nt(self.a32[i0 * nb:, j0 * nb:], prow, pcol, -1.0, 1.0)
        self.clamped = int(comm.allreduce_sum(t.tensor([self.clamped], dtype=t.int64, device=self.a32.device)).item()) if comm.on else self.clamped
        return self

    def _column_panel(self, k, prow, i0, j0, mc):
        """[mc, nb]: the tiles L_Jk for this rank's local block columns J > k, in local order."""
        t, g, nb, comm = self.torch, self.g, self.nb, self.comm
        pcol = t.zeros((mc, nb), dtype=t.float32, device=self.a32.device)
        if mc == 0:
            return pcol
        want = self.cols[j0:]  # global J of the local trailing block columns
        # rank (pr', pc) holds, inside ITS prow, the tiles I = pr' (mod Pr), I > k; it contributes those with I = pc (mod Pc)
        def contrib(prp):
            return [b for b in range(k + 1, self.NB) if b % g.Pr == prp and b % g.Pc == g.pc]
        mine = contrib(g.pr)
        cnt = max(1, max(len(contrib(p)) for p in range(g.Pr)))
        send = t.zeros((cnt * nb, nb), dtype=t.float32, device=self.a32.device)
        local_rows_after = self.rows[i0:]
        for s, b in enumerate(mine):
            li = local_rows_after.index(b)
            send[s * nb:(s + 1) * nb] = prow[li * nb:(li + 1) * nb]
        parts = comm.allgather(send, g.Pr, group=g.col_group)
        for prp in range(g.Pr):
            for s, b in enumerate(contrib(prp)):
                lj = want.index(b)
                pcol[lj * nb:(lj + 1) * nb] = parts[prp][s * nb:(s + 1) * nb]
        return pcol

    # ---- distributed triangular solves on a replicated block of right-hand-side rows ----
    def forward(self, b):
        """X = B L^-T; b: replicated [RHS, Np] float32, returns the same shape (replicated)."""
        t, g, nb, ops, comm = self.torch, self.g, self.nb, self.ops, self.comm
        x = t.zeros_like(b)
        for J in range(self.NB):
            part = t.zeros((b.shape[0], nb), dtype=t.float32, device=b.device)
            if g.pr == J % g.Pr:  # this process row owns block row J of L: tiles (J, K), K < J, K = pc (mod Pc)
                nk = self._local_pos(self.cols, J)
                if nk > 0:
                    iJ = self._local_pos(self.rows, J)
                    lrow = self.a32[iJ * nb:(iJ + 1) * nb, :nk * nb]  # [nb, nk nb], row-strided view
                    xsel = x[:, self.cidx[:nk * nb]].contiguous()
                    ops.gemm_nt(part, xsel, lrow, 1.0, 0.0)
                comm.reduce_sum(part, dst=g.owner(J, J), group=g.row_group)
            xj = t.zeros((b.shape[0], nb), dtype=t.float32, device=b.device)
            if g.rank == g.owner(J, J):
                rhs = (b[:, J * nb:(J + 1) * nb] - part).contiguous()
                ops.gemm_nt(xj, rhs, self.minv[J], 1.0, 0.0)  # rhs L_JJ^-T
            comm.bcast(xj, src=g.owner(J, J))
            x[:, J * nb:(J + 1) * nb] = xj
        return x

    def backward(self, b):
        """Y = B L^-1; b: replicated [RHS, Np] float32."""
        t, g, nb, ops, comm = self.torch, self.g, self.nb, self.ops, self.comm
        y = t.zeros_like(b)
        for J in range(self.NB - 1, -1, -1):
            part = t.zeros((b.shape[0], nb), dtype=t.float32, device=b.device)
            if g.pc == J % g.Pc:  # this process column owns block column J of L: tiles (K, J), K > J, K = pr (mod Pr)
                i0 = self._local_pos(self.rows, J + 1)
                if i0 < len(self.rows):
                    jJ = self._local_pos(self.cols, J)
                    lcol_t = self.a32[i0 * nb:, jJ * nb:(jJ + 1) * nb].t().contiguous()  # [nb, (K > J) nb]
                    ysel = y[:, self.ridx[i0 * nb:]].contiguous()
                    ops.gemm_nt(part, ysel, lcol_t, 1.0, 0.0)
                comm.reduce_sum(part, dst=g.owner(J, J), group=g.col_group)
            yj = t.zeros((b.shape[0], nb), dtype=t.float32, device=b.device)
            if g.rank == g.owner(J, J):
                rhs = (b[:, J * nb:(J + 1) * nb] - part).contiguous()
                ops.gemm_nt(yj, rhs, self.tinv[J], 1.0, 0.0)  # rhs L_JJ^-1 = rhs (L^-T)^T
            comm.bcast(yj, src=g.owner(J, J))
            y[:, J * nb:(J + 1) * nb] = yj
        return y

    def apply_inverse(self, b):
        return self.backward(self.forward(b))

    def matmul_a(self, p):
        """P (K + reg I) for a replicated block of rows p [RHS, Np] float64 (padding columns come back zero)."""
        t, ops, comm = self.torch, self.ops, self.comm
        q = t.zeros_like(p)
        if self.k64t.shape[0] > 0 and self.k64t.shape[1] > 0:
            part = t.zeros((p.shape[0], self.k64t.shape[0]), dtype=t.float64, device=p.device)
            ops.gemm_nt64(part, p[:, self.ridx].contiguous(), self.k64t, 1.0, 0.0)  # sum over this rank's block rows
            q[:, self.cidx] = part
        comm.allreduce_sum(q)
        q += self.reg * p
        q[:, self.n:] = 0.0
        return q

    # ---- alpha by preconditioned CG in float64 ----
    def solve(self, tol: float = 1e-10, max_iters: int = 60):
        t = self.torch
        R = self.RHS
        b = t.zeros((R, self.Np), dtype=t.float64, device=self.x.device)
        b[0, :self.n] = self.y
        xk = t.zeros_like(b)
        r = b.clone()
        bnorm = float(t.linalg.vector_norm(b[0]).item())
        z = self.apply_inverse(r.to(t.float32)).to(t.float64)
        z[:, self.n:] = 0.0
        p = z.clone()
        rz = float((r[0] * z[0]).sum().item())
        self.cg_iters, self.relres = 0, 0.0 if bnorm == 0.0 else 1.0
        for it in range(max_iters):
            if bnorm == 0.0:
                break
            q = self.matmul_a(p)
            pq = float((p[0] * q[0]).sum().item())
            if not pq > 0.0:
                break
            a = rz / pq
            xk[0] += a * p[0]
            r[0] -= a * q[0]
            self.cg_iters = it + 1
            self.relres = float(t.linalg.vector_norm(r[0]).item()) / bnorm
            if self.relres <= tol:
                break
            z = self.apply_inverse(r.to(t.float32)).to(t.float64)
            z[:, self.n:] = 0.0
            rz_new = float((r[0] * z[0]).sum().item())
            p[0] = z[0] + (rz_new / rz) * p[0]
            rz = rz_new
        self.alpha = xk[0, :self.n].clone()
        return self

    def fit(self):
        return self.build().factor().solve()

    # ---- predict: mean and diag variance of up to RHS test rows per call (replicated on every rank) ----
    def predict(self, x_test):
        t, ops = self.torch, self.ops
        xt_all = ops.to_device(np.asarray(x_test, dtype=np.float64))
        means, variances = [], []
        for s in range(0, xt_all.shape[0], self.RHS):
            xt = xt_all[s:s + self.RHS]
            m = xt.shape[0]
            k = t.zeros((self.RHS, self.Np), dtype=t.float64, device=xt.device)
            k[:m, :self.n] = ops.kernel(xt, self.x)
            ab = t.zeros((128, self.Np), dtype=t.float64, device=xt.device)  # alpha as row 0 of a 128-row block: K_td alpha is a GEMM
            ab[0, :self.n] = self.alpha
            mu = t.zeros((self.RHS, 128), dtype=t.float64, device=xt.device)
            ops.gemm_nt64(mu, k, ab, 1.0, 0.0)
            means.append(mu[:m, 0].clone())
            if ops.get != "nngp":  # the NTK-ensemble covariance needs both kernels distributed: not built in the 2-D form
                variances.append(t.full((m,), float("nan"), dtype=t.float64, device=xt.device))
                continue
            ktt, _ = ops.kernel_diag(xt)
            z0 = self.apply_inverse(k.to(t.float32)).to(t.float64)
            z0[:, self.n:] = 0.0
            r0 = k - self.matmul_a(z0)
            vr = self.forward(r0.to(t.float32)).to(t.float64)
            quad = (z0 * (k + r0)).sum(dim=1) + (vr * vr).sum(dim=1)
            variances.append(ktt - quad[:m])
        return t.cat(means).cpu().numpy(), t.cat(variances).cpu().numpy()
