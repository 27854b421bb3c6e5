"""``stax``-shaped front end of the closed-form NNGP/NTK kernel (reference call sites:
train.py:161-164, estimator.py:27-30, active/active_train.py:40-43).

    init_fn, apply_fn, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))
    k = kernel_fn(x1, x2, 'nngp')          # computed by the HIP kernel build (nngp_kernel_build)

Supported topology: Dense, (Relu, Dense)* -- what the reference builds; widths do not enter the
infinite-width kernel.  ``init_fn`` / ``apply_fn`` are the finite-width network in NumPy (NTK
parameterisation), kept for API shape; the reference never calls them.
"""
from __future__ import annotations

import collections
import ctypes
import math

import numpy as np

from . import _lib

Kernel = collections.namedtuple("Kernel", ["nngp", "ntk"])
_Layer = collections.namedtuple("_Layer", ["kind", "out_dim", "w_std", "b_std"])


def Dense(out_dim, W_std=1.0, b_std=None, parameterization="ntk"):
    if parameterization != "ntk":
        raise NotImplementedError("only the NTK parameterisation (the reference's default) is supported")
    return _Layer("dense", int(out_dim), float(W_std), 0.0 if b_std is None else float(b_std))


def Relu():
    return _Layer("relu", None, None, None)


class KernelFn:
    """kernel_fn(x1, x2=None, get=None): closed-form kernel of Dense,(Relu,Dense)* on the GPU."""

    def __init__(self, w_std, b_std):
        self.w_std = tuple(float(w) for w in w_std)
        self.b_std = tuple(float(b) for b in b_std)
        self.n_relu = len(self.w_std) - 1

    def _arch(self):
        return _lib.make_arch(self.w_std, self.b_std)

    def __call__(self, x1, x2=None, get=None, *, rows=None, as_numpy=True):
        import torch
        lib = _lib.load()
        dev = _lib.require_gpu()
        gets = ("nngp", "ntk") if get is None else ((get,) if isinstance(get, str) else tuple(get))
        for g in gets:
            if g not in ("nngp", "ntk"):
                raise ValueError("get must be 'nngp', 'ntk' or a tuple of them, got %r" % (get,))
        x1d = _lib.to_device_f64(x1, dev)
        if x1d.ndim != 2:
            raise ValueError("x1 must be [N, d]")
        x2d = None if x2 is None else _lib.to_device_f64(x2, dev)
        if x2d is not None and (x2d.ndim != 2 or x2d.shape[1] != x1d.shape[1]):
            raise ValueError("x2 must be [N2, d] with the same d as x1")
        n1, d = int(x1d.shape[0]), int(x1d.shape[1])
        n2 = n1 if x2d is None else int(x2d.shape[0])
        r0, r1 = (0, n1) if rows is None else (int(rows[0]), int(rows[1]))
        outs = {g: torch.empty((n1, n2), dtype=torch.float64, device=dev) for g in set(gets)}
        if n1 > 0 and n2 > 0 and r1 > r0:
            arch = self._arch()
            _lib.check(lib.nngp_kernel_build(_lib.ptr(x1d), n1, _lib.ptr(x2d), n2, d, ctypes.byref(arch),
                                             _lib.DTYPE_F64, _lib.ptr(outs.get("nngp")), _lib.ptr(outs.get("ntk")),
                                             n2, r0, r1, _lib.stream_ptr()))
        res = {g: (t[r0:r1] if rows is not None else t) for g, t in outs.items()}
        if as_numpy:
            res = {g: t.cpu().numpy() for g, t in res.items()}
        if get is None:
            return Kernel(res["nngp"], res["ntk"])
        if isinstance(get, str):
            return res[get]
        return collections.namedtuple("Kernel", gets)(*[res[g] for g in gets])


def serial(*layers):
    """Dense,(Relu,Dense)* -> (init_fn, apply_fn, kernel_fn)."""
    if not layers or any(not isinstance(l, _Layer) for l in layers):
        raise TypeError("serial() takes stax.Dense(...) / stax.Relu() layers")
    kinds = [l.kind for l in layers]
    ok = len(kinds) % 2 == 1 and all(k == ("dense" if i % 2 == 0 else "relu") for i, k in enumerate(kinds))
    if not ok:
        raise NotImplementedError("supported topology is Dense,(Relu,Dense)* -- got %s" % kinds)
    dense = [l for l in layers if l.kind == "dense"]
    kernel_fn = KernelFn([l.w_std for l in dense], [l.b_std for l in dense])

    def init_fn(rng, input_shape):
        gen = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)
        fan_in, params = int(input_shape[-1]), []
        for l in dense:
            params.append((gen.standard_normal((fan_in, l.out_dim)), gen.standard_normal((l.out_dim,))))
            fan_in = l.out_dim
        return tuple(input_shape[:-1]) + (fan_in,), params

    def apply_fn(params, x):
        h = np.asarray(x, dtype=np.float64)
        for i, (l, (w, b)) in enumerate(zip(dense, params)):
            h = l.w_std / math.sqrt(h.shape[-1]) * (h @ w) + l.b_std * b
            if i < len(dense) - 1:
                h = np.maximum(h, 0.0)
        return h

    return init_fn, apply_fn, kernel_fn
