"""ctypes front-end of oracle/libnngp_oracle.so (C float64/OpenMP restatement).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnngp_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nngp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnngp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_kernel_build.argtypes = [_dp, _i64, _dp, _i64, ctypes.c_int, ctypes.c_int, _dp, _dp,
                                             ctypes.c_int, _dp, _dp, _i64]
        _lib.oracle_potrf_lower.argtypes = [_dp, _i64, _i64]
        _lib.oracle_potrs_lower.argtypes = [_dp, _i64, _i64, _dp, _i64, _i64]
        _lib.oracle_fit.argtypes = [_dp, _dp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp,
                                    ctypes.c_int, ctypes.c_double, ctypes.c_int, _dp, _dp, _dp, _dp]
        _lib.oracle_predict_nngp.argtypes = [_dp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp,
                                             _dp, _dp, _dp, _i64, ctypes.c_int, _dp, _dp]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


GET = {"nngp": 1, "ntk": 2}


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_threads(n: int) -> int:
    lib().oracle_set_threads(int(n))
    return num_threads()


def kernel_build(x1, x2, get, w_std, b_std):
    x1 = _c(x1)
    x2c = None if x2 is None else _c(x2)
    n1, d = x1.shape
    n2 = n1 if x2c is None else x2c.shape[0]
    w, b = _c(w_std), _c(b_std)
    out = np.empty((n1, n2))
    g = GET[get]
    rc = lib().oracle_kernel_build(_p(x1), n1, _p(x2c), n2, d, len(w), _p(w), _p(b), g,
                                   _p(out) if g == 1 else None, _p(out) if g == 2 else None, n2)
    assert rc == 0
    return out


def potrf_lower(a):
    a = _c(a).copy()
    info = lib().oracle_potrf_lower(_p(a), a.shape[0], a.shape[1])
    return np.tril(a), info


def fit(x, y, w_std, b_std, get="nngp", diag_reg=1e-3, absolute=False):
    x = _c(x)
    n, d = x.shape
    y = _c(y).reshape(n, -1)
    w, b = _c(w_std), _c(b_std)
    l_out = np.empty((n, n))
    alpha = np.empty_like(y)
    reg = ctypes.c_double(0.0)
    stages = np.zeros(3)
    rc = lib().oracle_fit(_p(x), _p(y), n, d, y.shape[1], len(w), _p(w), _p(b), GET[get], diag_reg,
                          int(absolute), _p(l_out), _p(alpha), ctypes.cast(ctypes.byref(reg), _dp), _p(stages))
    if rc != 0:
        raise RuntimeError("oracle_fit failed: %d" % rc)
    return {"x": x, "L": l_out, "alpha": alpha, "reg": reg.value, "w": w, "b": b, "stage_sec": stages}


def predict_nngp(model, x_test, cov_mode=1):
    x_test = _c(x_test)
    m = x_test.shape[0]
    x = model["x"]
    n, d = x.shape
    ny = model["alpha"].shape[1]
    mean = np.empty((m, ny))
    out = np.empty((m, m) if cov_mode == 2 else (max(m, 1),))
    rc = lib().oracle_predict_nngp(_p(x), n, d, ny, len(model["w"]), _p(model["w"]), _p(model["b"]),
                                   _p(model["L"]), _p(model["alpha"]), _p(x_test), m, cov_mode, _p(mean), _p(out))
    assert rc == 0
    return (mean, out) if cov_mode else mean
