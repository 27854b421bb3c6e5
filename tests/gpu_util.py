"""Helpers for the -m gpu parity tests: thin torch <-> C-ABI glue (everything goes through libnngp_hip.so)."""
import ctypes

import numpy as np
import torch

from nngp_src_amd import _lib


def dev():
    return torch.device("cuda", 0)


def kernel_build(x1, x2, w_std, b_std, get=("nngp", "ntk"), rows=None, dtype=torch.float64, ld=None, knobs=False):
    lib = _lib.load(knobs=knobs)
    x1d = _lib.to_device_f64(x1, dev())
    x2d = None if x2 is None else _lib.to_device_f64(x2, dev())
    n1, d = x1d.shape
    n2 = n1 if x2d is None else x2d.shape[0]
    ld = n2 if ld is None else ld
    outs = {g: torch.full((n1, ld), float("nan"), dtype=dtype, device=dev()) for g in get}
    arch = _lib.make_arch(w_std, b_std)
    r0, r1 = (0, n1) if rows is None else rows
    _lib.check(lib.nngp_kernel_build(_lib.ptr(x1d), n1, _lib.ptr(x2d), n2, d, ctypes.byref(arch),
                                     _lib.DTYPE_F64 if dtype == torch.float64 else _lib.DTYPE_F32,
                                     _lib.ptr(outs.get("nngp")), _lib.ptr(outs.get("ntk")), ld, r0, r1, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return {g: t.cpu().numpy() for g, t in outs.items()}


def gemm_nt(c, a, b, alpha, beta, lower_only=False):
    lib = _lib.load()
    m, k = a.shape
    n = b.shape[0]
    _lib.check(lib.nngp_gemm_nt_f32(_lib.ptr(c), c.stride(0), _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0),
                                    m, n, k, alpha, beta, int(lower_only), _lib.stream_ptr()))
    torch.cuda.synchronize()


def gemm_nt_h3(c, a, b, alpha, beta, scale, lower_only=False):
    lib = _lib.load()
    m, k = a.shape
    n = b.shape[0]
    _lib.check(lib.nngp_gemm_nt_h3(_lib.ptr(c), c.stride(0), _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0),
                                   m, n, k, alpha, beta, scale, int(lower_only), _lib.stream_ptr()))
    torch.cuda.synchronize()


def gemm_nt_f64(c, cin, a, b, alpha, beta):
    lib = _lib.load()
    m, k = a.shape
    n = b.shape[0]
    _lib.check(lib.nngp_gemm_nt_f64(_lib.ptr(c), c.stride(0), _lib.ptr(cin), 0 if cin is None else cin.stride(0), _lib.ptr(a),
                                    a.stride(0), _lib.ptr(b), b.stride(0), m, n, k, alpha, beta, _lib.stream_ptr()))
    torch.cuda.synchronize()


def gemm_nt_i8s(c, cin, a, b, alpha, beta, slices_a=5, slices_b=5, cut=4):
    lib = _lib.load()
    m, k = a.shape
    n = b.shape[0]
    _lib.check(lib.nngp_gemm_nt_i8s(_lib.ptr(c), c.stride(0), _lib.ptr(cin), 0 if cin is None else cin.stride(0), _lib.ptr(a),
                                    a.stride(0), _lib.ptr(b), b.stride(0), m, n, k, alpha, beta, slices_a, slices_b, cut,
                                    _lib.stream_ptr()))
    torch.cuda.synchronize()


def potrf(a):
    """a: [n, n] float32 cuda tensor (lower triangle used); returns (dinv [n/128,128,128], clamped)."""
    lib = _lib.load()
    n = a.shape[0]
    dinv = torch.full((n // 128, 128, 128), float("nan"), dtype=torch.float32, device=a.device)
    clamped = torch.zeros(1, dtype=torch.int32, device=a.device)
    _lib.check(lib.nngp_potrf_f32(_lib.ptr(a), n, a.stride(0), _lib.ptr(dinv), _lib.ptr(clamped), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return dinv, int(clamped.item())


def trsm(b, l, dinv):
    lib = _lib.load()
    _lib.check(lib.nngp_trsm_rlt_f32(_lib.ptr(b), b.stride(0), b.shape[0], _lib.ptr(l), l.stride(0), _lib.ptr(dinv),
                                     l.shape[0], _lib.stream_ptr()))
    torch.cuda.synchronize()


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def mean_gate(mean, ref):
    """SURVEY.md 8d parity gate: |d mu|_2/|mu|_2 <= 1e-4 and |d mu_i| <= 1e-4 max(1, |mu_i|)."""
    mean, ref = np.asarray(mean).ravel(), np.asarray(ref).ravel()
    return rel_l2(mean, ref), float(np.max(np.abs(mean - ref) / np.maximum(1.0, np.abs(ref))))
