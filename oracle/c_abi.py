"""ctypes front-end of oracle/libnngp_cpu.so: the C ABI of include/nngp_hip.h compiled for the host (nngp_cpu_abi.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py; never
from the product package.  The prototypes are the product's own (nngp_src_amd._lib.bind_prototypes): the host build and the
HIP build sit behind one interface, with host pointers here where the HIP library takes device pointers.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnngp_cpu.so")
_lib = None

COV = {None: 0, "none": 0, "diag": 1, "full": 2, True: 2, False: 0}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("nngp_cpu_abi.c", "nngp_oracle.c", "cpu_abi_glue.cpp")]
    srcs += [os.path.join(_HERE, "..", "nngp-src_amd", "csrc", "encoder.cpp"), os.path.join(_HERE, "..", "include", "nngp_hip.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnngp_cpu.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        from nngp_src_amd import _lib as proto  # prototypes only: no HIP library is loaded by this import
        _lib = proto.bind_prototypes(ctypes.CDLL(LIB_PATH))
    return _lib


def num_threads() -> int:
    f = lib().oracle_num_threads
    f.restype = ctypes.c_int
    return f()


def set_threads(n: int) -> int:
    lib().oracle_set_threads(int(n))
    return num_threads()


def _check(rc):
    if rc != 0:
        raise RuntimeError("libnngp_cpu: rc=%d: %s" % (rc, (lib().nngp_last_error() or b"").decode("utf-8", "replace")))


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CpuModel:
    """nngp_model_* of the host build: the call sequence of nngp_src_amd.model.GPModel on NumPy arrays."""

    def __init__(self, n_cap, d, w_std, b_std, get="nngp", diag_reg=1e-3, diag_reg_absolute_scale=False, ny=1, m_cap=0):
        from nngp_src_amd import _lib as proto
        self.lib = lib()
        self.arch = proto.make_arch(w_std, b_std)
        self.d, self.ny, self.get = int(d), int(ny), get
        self.handle = ctypes.c_void_p()
        _check(self.lib.nngp_model_create(ctypes.byref(self.handle), int(n_cap), int(m_cap), self.d, self.ny,
                                          ctypes.byref(self.arch), {"nngp": 1, "ntk": 2}[get], float(diag_reg),
                                          1 if diag_reg_absolute_scale else 0))
        self.n = 0

    def __del__(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.nngp_model_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def fit(self, x, y):
        x, y = _c(x), _c(y).reshape(len(x), -1)
        self.n = x.shape[0]
        _check(self.lib.nngp_model_fit(self.handle, _p(x), _p(y), self.n, None))
        return self

    def set_train(self, x, y):
        x, y = _c(x), _c(y).reshape(len(x), -1)
        self.n = x.shape[0]
        _check(self.lib.nngp_model_set_train(self.handle, _p(x), _p(y), self.n, None))

    def build_rows(self, r0, r1):
        _check(self.lib.nngp_model_build_rows(self.handle, int(r0), int(r1), None))

    def factor(self):
        _check(self.lib.nngp_model_factor(self.handle, None))

    def solve(self):
        _check(self.lib.nngp_model_solve(self.handle, 0, 0.0, None))

    def alpha(self):
        out = np.empty((self.n, self.ny))
        _check(self.lib.nngp_model_alpha(self.handle, _p(out), None))
        return out

    def info(self):
        from nngp_src_amd import _lib as proto
        fi = proto.NngpFitInfo()
        _check(self.lib.nngp_model_info(self.handle, ctypes.byref(fi)))
        return {k: getattr(fi, k) for k, _ in fi._fields_}

    def predict(self, x_test=None, cov="diag"):
        mode = COV[cov]
        xt = None if x_test is None else _c(x_test)
        mt = self.n if xt is None else xt.shape[0]
        mean = np.empty((mt, self.ny))
        out = None if mode == 0 else (np.empty(mt) if mode == 1 else np.empty((mt, mt)))
        _check(self.lib.nngp_model_predict(self.handle, _p(xt), mt, mode, _p(mean), _p(out), None))
        return mean if mode == 0 else (mean, out)


def kernel_build(x1, x2, get, w_std, b_std, rows=None, dtype=np.float64):
    """nngp_kernel_build of the host build: rows [r0, r1) of the n1 x n2 kernel (NaN elsewhere)."""
    from nngp_src_amd import _lib as proto
    x1 = _c(x1)
    x2c = None if x2 is None else _c(x2)
    n1, d = x1.shape
    n2 = n1 if x2c is None else x2c.shape[0]
    arch = proto.make_arch(w_std, b_std)
    r0, r1 = (0, n1) if rows is None else rows
    out = np.full((n1, n2), np.nan, dtype=dtype)
    args = [_p(out) if get == "nngp" else None, _p(out) if get == "ntk" else None]
    _check(lib().nngp_kernel_build(_p(x1), n1, _p(x2c), n2, d, ctypes.byref(arch), 1 if dtype == np.float64 else 0,
                                   args[0], args[1], n2, r0, r1, None))
    return out


def pool_select(mean, var, count, biased=False, seed=10):
    """nngp_pool_select of the host build: indices of the pool queries the active-learning loop moves to the training set."""
    mean = _c(np.asarray(mean).reshape(len(var), -1))
    var = _c(var)
    idx = np.empty((int(count),), dtype=np.int64)
    _check(lib().nngp_pool_select(_p(mean), mean.shape[0], mean.shape[1], _p(var), int(count), int(bool(biased)), int(seed),
                                  idx.ctypes.data_as(ctypes.c_void_p), None))
    return idx
