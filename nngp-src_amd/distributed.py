"""Row-block data parallelism for the kernel build (the ``nt.batch(device_count=...)`` slot).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU
tests).  Row i of K depends only on x_i and X, so rank g builds the contiguous block
``[g*chunk, (g+1)*chunk)`` and ONE all-gather gives every rank the full float64 kernel for the replicated
Cholesky.  No other collective touches the data path.
"""
from __future__ import annotations

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world_size() -> int:
    dist = _dist()
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    dist = _dist()
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def row_chunk(n: int, world: int) -> int:
    return (n + world - 1) // world


def row_partition(n: int, world: int, r: int):
    """Rows [r0, r1) owned by rank r: equal chunks of ceil(n/world), the tail ranks may be short or empty."""
    c = row_chunk(n, world)
    return min(r * c, n), min((r + 1) * c, n)


class NativeComm:
    """RCCL communicator owned by libnngp_hip.so (``nngp_comm_*``, include/nngp_hip.h section e).

    ``torch.distributed`` only carries the 128-byte unique id from rank 0 to the others; the all-gather itself is the
    library's ``nngp_allgather_rows`` (ncclAllGather, in place) on the caller's HIP stream.  One rank per GPU -- RCCL
    refuses two ranks on one device, so the gloo rehearsals on a single GPU cannot use it."""

    def __init__(self, group=None):
        """Failure-symmetric rendezvous: every rank runs the SAME sequence of ``torch.distributed`` collectives whether or not
        a step failed on it, and either every rank ends with a communicator or every rank raises ``NngpError`` -- a rank never
        leaves the others inside a collective it skipped.  (1) each rank binds librccl, all agree (all_reduce MIN); (2) rank 0
        creates the id and ALWAYS broadcasts 129 bytes = status + id; (3) ``ncclCommInitRank`` on every rank, all agree."""
        import ctypes
        import torch
        from . import _lib
        dist = _dist()
        self.lib = _lib.load()
        self.handle = None
        self.world, self.rank = world_size(), rank()
        multi = self.world > 1
        dev = (torch.device("cuda", torch.cuda.current_device()) if multi and dist.get_backend(group) == "nccl"
               else torch.device("cpu"))

        def all_ok(ok: bool) -> bool:
            if not multi:
                return ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        self.library = (self.lib.nngp_comm_library() or b"").decode()
        mine = "" if self.library else (self.lib.nngp_last_error() or b"librccl not found").decode()
        if not all_ok(bool(self.library)):
            raise _lib.NngpError("RCCL could not be bound on every rank" + (": " + mine if mine else " (another rank failed)"))
        ident = ctypes.create_string_buffer(128)
        status, mine = 0, ""
        if self.rank == 0 and self.lib.nngp_comm_unique_id(ident) != 0:
            status, mine = 1, (self.lib.nngp_last_error() or b"").decode()
        if multi:
            t = torch.tensor([status] + list(ident.raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0, group=group)
            raw = bytes(t.cpu().tolist())
            status, ident = raw[0], ctypes.create_string_buffer(raw[1:], 128)
        if status != 0:
            raise _lib.NngpError("ncclGetUniqueId failed on rank 0" + (": " + mine if mine else ""))
        handle = ctypes.c_void_p()
        rc = self.lib.nngp_comm_create(ctypes.byref(handle), ident, self.world, self.rank)
        mine = "" if rc == 0 else (self.lib.nngp_last_error() or b"").decode()
        if not all_ok(rc == 0):
            if rc == 0:
                self.lib.nngp_comm_destroy(handle)
            raise _lib.NngpError("ncclCommInitRank did not succeed on every rank" + (": " + mine if mine else " (another rank failed)"))
        self.handle = handle

    def allgather_rows(self, buf, n: int):
        from . import _lib
        import torch
        if not buf.is_contiguous() or buf.shape[0] < self.world * row_chunk(n, self.world):
            raise ValueError("all-gather buffer must be contiguous with >= world * chunk rows")
        dt = {torch.float64: _lib.DTYPE_F64, torch.float32: _lib.DTYPE_F32}[buf.dtype]
        _lib.check(self.lib.nngp_allgather_rows(_lib.ptr(buf), int(n), int(buf.stride(0)), dt, self.handle, _lib.stream_ptr()),
                   self.lib)
        return buf

    def bcast(self, t, root: int):
        from . import _lib
        import torch
        dt = {torch.float64: _lib.DTYPE_F64, torch.float32: _lib.DTYPE_F32}[t.dtype]
        _lib.check(self.lib.nngp_bcast(_lib.ptr(t), t.numel(), dt, int(root), self.handle, _lib.stream_ptr()), self.lib)
        return t

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nngp_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_rows(buf, n: int, group=None, comm: "NativeComm" = None):
    """All-gather the row blocks of ``buf`` ([rows >= world*chunk, ld] torch tensor) IN PLACE.

    Every rank must have written its own rows ``row_partition(n, world, rank)``; on return rows [0, n) are
    complete on every rank.  Sends exactly chunk*ld elements per rank (the short tail block is padded by
    whatever the buffer holds -- rows >= n are never read); the send block is the rank's own slice of the receive
    buffer, so nothing is copied.  ``comm``: the library's own RCCL communicator (``NativeComm``); otherwise
    ``torch.distributed`` (nccl = RCCL on ROCm; gloo in the rehearsals, staged through the host).
    """
    import torch
    dist = _dist()
    world, r = world_size(), rank()
    if world == 1:
        return buf
    c = row_chunk(n, world)
    if buf.shape[0] < world * c:
        raise ValueError("all-gather buffer has %d rows, needs %d" % (buf.shape[0], world * c))
    if comm is not None:
        return comm.allgather_rows(buf, n)
    out = buf[: world * c]
    mine = out[r * c:(r + 1) * c]
    if buf.is_cuda and dist.get_backend(group) != "nccl":
        # rehearsal path (gloo with device buffers): stage through the host
        host = mine.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        for g, p in enumerate(parts):
            if g != r:
                out[g * c:(g + 1) * c].copy_(p)
    elif out.is_contiguous() and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=group)  # in place: send = own slice of recv
    elif out.is_contiguous():
        dist.all_gather_into_tensor(out.view(-1), mine.reshape(-1).clone(), group=group)  # gloo: no aliasing
    else:  # pragma: no cover
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.contiguous(), group=group)
        for g, p in enumerate(parts):
            out[g * c:(g + 1) * c].copy_(p)
    return buf


def sharded_kernel(kernel_fn, x1, x2=None, get="nngp", group=None):
    """kernel_fn over row blocks of x1 on every rank + one all-gather; returns the full [N1, N2] array."""
    import torch
    world, r = world_size(), rank()
    n1 = int(np.shape(x1)[0])
    n2 = n1 if x2 is None else int(np.shape(x2)[0])
    if world == 1:
        return kernel_fn(x1, x2, get)
    if not isinstance(get, str):
        raise NotImplementedError("sharded_kernel: one kernel per call")
    c = row_chunk(n1, world)
    r0, r1 = row_partition(n1, world, r)
    backend = _dist().get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((world * c, n2), dtype=torch.float64, device=dev)
    if r1 > r0:
        blk = kernel_fn(x1, x2, get, rows=(r0, r1))
        buf[r0:r1] = torch.as_tensor(np.asarray(blk), dtype=torch.float64).to(dev) if not isinstance(blk, torch.Tensor) else blk
    allgather_rows(buf, n1, group)
    return buf[:n1].cpu().numpy()


def panel_width(np_: int) -> int:
    """Block-column width of the distributed Cholesky (multiple of 128)."""
    return 1024 if np_ >= 8192 else (512 if np_ >= 2048 else 128)


class _Done:
    def wait(self):
        return True


def _broadcast(t, src, group=None, async_op=False):
    """Broadcast a contiguous device tensor and return a handle with .wait().  nccl (= RCCL): asynchronous on the
    communicator's stream; gloo rehearsals stage through the host synchronously."""
    dist = _dist()
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        if rank() != src:
            t.copy_(h)
        return _Done()
    work = dist.broadcast(t, src=src, group=group, async_op=async_op)
    return work if async_op else _Done()


class _NativeBcast:
    """Panel broadcasts through the library's own RCCL communicator (``nngp_bcast``) on a side stream: the transfer of block
    column k+1 over xGMI runs beside the trailing updates of block column k on the compute stream.  ``wait()`` makes the
    compute stream wait for the transfer (stream-ordered: no host synchronisation)."""

    def __init__(self, comm):
        import torch
        self.comm, self.torch = comm, torch
        self.stream = torch.cuda.Stream()

    def start(self, buf, src):
        t = self.torch
        ready = t.cuda.Event()
        ready.record(t.cuda.current_stream())       # the owner's pack copies (and every rank's last use of this staging buffer)
        self.stream.wait_event(ready)
        with t.cuda.stream(self.stream):
            self.comm.bcast(buf, src)               # ncclBroadcast in place on the side stream
            done = t.cuda.Event()
            done.record(self.stream)
        outer = self

        class _Handle:
            def wait(self_inner):
                outer.torch.cuda.current_stream().wait_event(done)
                return True
        return _Handle()


def distributed_factor(model, group=None, nb: int = None, comm: "NativeComm" = None):
    """Right-looking float32 Cholesky with the block columns dealt cyclically to the ranks (SURVEY.md 8f row N4, 1-D form).

    Every rank keeps a full copy of the factor buffer.  Block column k is final on its owner (k mod world) once the
    owner has applied columns < k to it; the owner factors it (diagonal block + rows below) and broadcasts it together
    with its inverted 128-blocks (one RCCL broadcast of (n - o) * w + w * 128 floats); every rank stores it in place
    and applies it to the block columns it owns.  Each rank therefore does 1/world of the O(N^3) trailing updates and
    ends with the complete factor -- the replicated triangular solves that follow need no further exchange.

    Look-ahead: after receiving column k, the owner of column k+1 updates and factors THAT column first and starts
    its broadcast; all ranks post the (asynchronous) receive before applying column k to the rest of their columns,
    so the transfer of column k+1 over xGMI overlaps with the trailing updates of column k.
    ``comm``: the library's RCCL communicator (``NativeComm``): the broadcasts then go through ``nngp_bcast`` on a side
    stream instead of ``torch.distributed``.
    """
    import torch
    world, r = world_size(), rank()
    model.factor_begin()
    a32, dinv = model.factor_buffers()
    np_ = a32.shape[0]
    w = nb or panel_width(np_)
    ncols = (np_ + w - 1) // w
    owned = [j for j in range(ncols) if j % world == r]
    width = lambda j: min(w, np_ - j * w)
    stages = [torch.empty(((np_ * w) + w * 128,), dtype=torch.float32, device=a32.device) for _ in range(2)] if world > 1 else None
    # The panels travel by torch.distributed's broadcast.  NNGP_BCAST=native: by nngp_bcast on a side stream of the library's own
    # communicator -- opt-in until one run on two or more GPUs has checked that path against the single-GPU factor (it has only
    # ever run as a one-rank self-broadcast: no multi-GPU box was available to rounds 1-4; tests/test_gpu_distributed.py holds the
    # two-GPU test that will)
    import os
    native = _NativeBcast(comm) if (comm is not None and world > 1 and os.environ.get("NNGP_BCAST", "torch") == "native") else None

    def views(k):
        o, wk = k * w, width(k)
        rows = np_ - o
        buf = stages[k & 1][: rows * wk + wk * 128]
        return o, wk, buf[: rows * wk].view(rows, wk), buf[rows * wk:].view(wk // 128, 128, 128), buf

    def factor_and_send(k):
        """Owner: factor block column k and pack it; everyone: post the broadcast.  Returns the handle."""
        o, wk, pan, inv, buf = views(k)
        if k % world == r:
            model.factor_panel(o, wk)
            pan.copy_(a32[o:, o:o + wk])
            inv.copy_(dinv[o // 128:(o + wk) // 128])
        if native is not None:
            return native.start(buf, k % world)
        return _broadcast(buf, k % world, group, async_op=True)

    def receive(k, handle):
        handle.wait()
        if k % world != r:
            o, wk, pan, inv, _ = views(k)
            a32[o:, o:o + wk].copy_(pan)
            dinv[o // 128:(o + wk) // 128].copy_(inv)

    if world == 1:
        for k in range(ncols):
            model.factor_panel(k * w, width(k))
            model.factor_update_cols(k * w, width(k), [j * w for j in range(k + 1, ncols)], w)
        model.factor_end()
        return model

    handle = factor_and_send(0)
    for k in range(ncols):
        receive(k, handle)
        o, wk = k * w, width(k)
        nxt = k + 1
        if nxt < ncols:
            if nxt % world == r:
                model.factor_update(o, wk, nxt * w, width(nxt))  # column k+1 first: it is on the critical path
            handle = factor_and_send(nxt)
        # the rest of this rank's block columns: four to a launch (nngp_model_factor_update_cols)
        model.factor_update_cols(o, wk, [j * w for j in owned if j > nxt], w)
    model.factor_end()
    return model


def _any_rank(flag: bool, group=None) -> bool:
    """Logical OR of a host flag over the ranks."""
    import torch
    dist = _dist()
    if world_size() == 1:
        return bool(flag)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def sharded_fit(model, x, y, group=None, timings=None, distributed_cholesky=True, comm=None):
    """GPModel fit with the kernel build sharded over ranks: build own rows -> all-gather -> replicated
    factor + solve.  The model must have been created with n_cap >= world * ceil(n / world)."""
    import torch
    world, r = world_size(), rank()
    model.set_train(x, y)
    n = model.n
    r0, r1 = row_partition(n, world, r)
    ev = lambda: _event()
    t0 = ev()
    if world == 1:
        model.build_rows(0, n)
    else:
        model.build_rows(r0, r1)
    t1 = ev()
    if world > 1:
        buf, _ = model.kernel_buffer(all_rows=True)
        allgather_rows(buf, n, group, comm)
    t2 = ev()
    if world > 1 and distributed_cholesky:
        distributed_factor(model, group, comm=comm)
        # float32 breakdown (clamped pivots on any rank's block columns: cond(K + reg I) * eps32 >> 1): every rank holds
        # the whole float64 kernel, so all of them redo the factor on their own GPU, where nngp_model_factor raises the
        # preconditioner shift until the factorisation goes through (one 4-byte status reduction per fit)
        if _any_rank(model.info()["clamped_pivots"] > 0, group):
            model.factor()
    else:
        model.factor()
    t3 = ev()
    model.solve()
    t4 = ev()
    if timings is not None:
        torch.cuda.synchronize()
        timings.update(build_ms=t0.elapsed_time(t1), allgather_ms=t1.elapsed_time(t2), factor_ms=t2.elapsed_time(t3),
                       solve_ms=t3.elapsed_time(t4))
    return model


def _event():
    import torch
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    return e
