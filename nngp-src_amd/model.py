"""GPModel: Python owner of one ``nngp_model`` handle (C ABI, include/nngp_hip.h).

One object = one exact GP posterior for one ``get`` ('nngp' or 'ntk'): the float64 train-train kernel,
its float32 MFMA Cholesky factor and alpha = (K + reg I)^-1 Y, all resident in HBM.  This is what the
closure returned by ``nt.predict.gradient_descent_mse_ensemble`` caches in the reference
(train.py:171-172).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

_GET = {"nngp": _lib.GET_NNGP, "ntk": _lib.GET_NTK}
_COV = {False: _lib.COV_NONE, None: _lib.COV_NONE, "none": _lib.COV_NONE, "diag": _lib.COV_DIAG,
        True: _lib.COV_FULL, "full": _lib.COV_FULL}


class GPModel:
    def __init__(self, n_cap: int, d: int, w_std, b_std, get: str = "nngp", diag_reg: float = 1e-3,
                 diag_reg_absolute_scale: bool = False, ny: int = 1, m_cap: int = 0, knobs: bool = False):
        if get not in _GET:
            raise ValueError("get must be 'nngp' or 'ntk', got %r" % (get,))
        self.lib = _lib.load(knobs)  # knobs=True: the timing-knob build (A/B tests and scripts/ only)
        self.device = _lib.require_gpu()
        self.get, self.d, self.ny, self.n_cap = get, int(d), int(ny), int(n_cap)
        self.arch = _lib.make_arch(w_std, b_std)
        self.handle = ctypes.c_void_p()
        self._check(self.lib.nngp_model_create(ctypes.byref(self.handle), int(n_cap), int(m_cap), int(d), int(ny),
                                              ctypes.byref(self.arch), _GET[get], float(diag_reg),
                                              int(bool(diag_reg_absolute_scale))))
        if m_cap > 0:  # predict-side workspace now, not inside the first predict (SURVEY 8b ownership rule)
            self._check(self.lib.nngp_model_reserve(self.handle, int(m_cap), _lib.COV_DIAG))
        self.n = 0
        self._diag_reg, self._absolute = float(diag_reg), bool(diag_reg_absolute_scale)
        self._keep = []  # device tensors that must outlive asynchronous work

    def _check(self, rc: int):
        _lib.check(rc, self.lib)

    def debug_set(self, key: int, value: int):
        """Timing-experiment switch; only a model created with ``knobs=True`` has it (libnngp_hip_knobs.so)."""
        if not hasattr(self.lib, "nngp_debug_set"):
            raise _lib.NngpError("the product library has no nngp_debug_set: create the model with knobs=True")
        self._check(self.lib.nngp_debug_set(int(key), int(value)))

    # ---- lifetime ----
    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.nngp_model_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stages (bench.py times them separately; the multi-GPU path shards build_rows) ----
    def set_train(self, x, y):
        xd = _lib.to_device_f64(x, self.device)
        yd = _lib.to_device_f64(y, self.device).reshape(xd.shape[0], -1)
        if xd.ndim != 2 or xd.shape[1] != self.d:
            raise ValueError("x_train must be [N, %d], got %s" % (self.d, tuple(xd.shape)))
        if yd.shape[1] != self.ny:
            raise ValueError("y_train must have %d column(s), got %s" % (self.ny, tuple(yd.shape)))
        if int(xd.shape[0]) > self.n_cap or xd.shape[0] == 0:
            raise ValueError("x_train has %d rows; the model was created for 1..%d" % (xd.shape[0], self.n_cap))
        self.n = int(xd.shape[0])
        self._keep = [xd, yd]
        self._check(self.lib.nngp_model_set_train(self.handle, _lib.ptr(xd), _lib.ptr(yd), self.n, _lib.stream_ptr()))

    def append(self, x_new, y_new, solve: bool = True):
        """Add training rows to a fitted model: their kernel rows are built and the factor is extended in place
        (``nngp_model_append``: cost ~ b N^2 instead of a full N^3/3 refit), then alpha is re-solved."""
        xd = _lib.to_device_f64(x_new, self.device)
        yd = _lib.to_device_f64(y_new, self.device).reshape(xd.shape[0], -1)
        if xd.ndim != 2 or xd.shape[1] != self.d or yd.shape[1] != self.ny:
            raise ValueError("append: x_new must be [b, %d] and y_new [b, %d]" % (self.d, self.ny))
        b = int(xd.shape[0])
        if b == 0:
            return self
        if self.n + b > self.n_cap:
            raise ValueError("append: %d + %d rows exceed the capacity %d" % (self.n, b, self.n_cap))
        self._check(self.lib.nngp_model_append(self.handle, _lib.ptr(xd), _lib.ptr(yd), b, _lib.stream_ptr()))
        import torch
        self._keep = [torch.cat([self._keep[0], xd]), torch.cat([self._keep[1], yd])]  # what save() writes
        self.n += b
        if solve:
            self.solve()
        return self

    def build_rows(self, row_begin: int = 0, row_end: int = None):
        row_end = self.n if row_end is None else row_end
        self._check(self.lib.nngp_model_build_rows(self.handle, int(row_begin), int(row_end), _lib.stream_ptr()))

    def factor(self):
        self._check(self.lib.nngp_model_factor(self.handle, _lib.stream_ptr()))

    # block-column pieces of factor() for the multi-GPU Cholesky (distributed.distributed_factor)
    def factor_begin(self):
        self._check(self.lib.nngp_model_factor_begin(self.handle, _lib.stream_ptr()))

    def factor_panel(self, col0: int, width: int):
        self._check(self.lib.nngp_model_factor_panel(self.handle, int(col0), int(width), _lib.stream_ptr()))

    def factor_update(self, panel_col0: int, panel_width: int, col0: int, width: int):
        self._check(self.lib.nngp_model_factor_update(self.handle, int(panel_col0), int(panel_width), int(col0), int(width),
                                                     _lib.stream_ptr()))

    def factor_update_cols(self, panel_col0: int, panel_width: int, cols, width: int):
        """Apply block column [panel_col0, +panel_width) to several target block columns (first rows / columns ``cols``, ascending):
        the block columns a rank owns go out four to a split-float16 launch (nngp_model_factor_update_cols)."""
        cols = [int(c) for c in cols]
        if not cols:
            return
        arr = (ctypes.c_int64 * len(cols))(*cols)
        self._check(self.lib.nngp_model_factor_update_cols(self.handle, int(panel_col0), int(panel_width), arr, len(cols), int(width),
                                                          _lib.stream_ptr()))

    def factor_end(self):
        self._check(self.lib.nngp_model_factor_end(self.handle, _lib.stream_ptr()))

    def factor_buffers(self):
        """(a32 [np, np] float32 view, dinv [np/128, 128, 128] float32 view) of the library-owned factor buffers."""
        a, ld, d = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_void_p()
        self._check(self.lib.nngp_model_factor_buffers(self.handle, ctypes.byref(a), ctypes.byref(ld), ctypes.byref(d)))
        np_ = self.info()["n_padded"]  # rows/columns in use; the row stride ld is the padded capacity
        a32 = _wrap_device(a.value, np_ * ld.value, self.device, "<f4").view(np_, ld.value)[:, :np_]
        dinv = _wrap_device(d.value, (np_ // 128) * 128 * 128, self.device, "<f4").view(np_ // 128, 128, 128)
        return a32, dinv

    def apply_factor(self, b, both_halves: bool = False):
        """b [rows, n] float32 CUDA tensor (contiguous), overwritten with ``b L^-T`` (``both_halves``: ``b (L L^T)^-1``) through the
        blocked solves the posterior takes for a block of that many rows (nngp_model_apply_factor)."""
        import torch
        assert b.is_cuda and b.dtype == torch.float32 and b.is_contiguous() and b.dim() == 2
        self._check(self.lib.nngp_model_apply_factor(self.handle, b.data_ptr(), b.shape[0], 1 if both_halves else 0, _lib.stream_ptr()))
        return b

    def reserve(self, rows: int, cov="diag"):
        """Allocate now what a predict of up to ``rows`` test rows with this covariance mode would allocate on first use."""
        self._check(self.lib.nngp_model_reserve(self.handle, int(rows), _COV[cov]))
        return self

    def solve(self, max_iters: int = 0, tol: float = 0.0):
        self._check(self.lib.nngp_model_solve(self.handle, int(max_iters), float(tol), _lib.stream_ptr()))

    def set_refine(self, sweeps: int):
        """Precision level of the posterior covariance.  0 = float32 solve only (~1e-2 at N = 32768); 1 (default) = one
        float64 residual: second-order formula at the float32 solution plus the preconditioned estimate of its remainder
        (diag: ~2e-6 at N = 32768; full covariance and NTK run at level 2); L >= 2 = L-1 correction sweeps plus the
        second-order formula.  Levels >= 1 continue by per-row CG when the float32 factor is a weak preconditioner."""
        self._check(self.lib.nngp_model_set_refine(self.handle, int(sweeps)))
        return self

    def prepare_serving(self):
        """Build the explicit float64 inverse of K + reg I once (``nngp_model_prepare_serving``): later ``predict`` calls
        cost one float64 product instead of blocked solves and correction sweeps.  Dropped by fit / append."""
        self._check(self.lib.nngp_model_prepare_serving(self.handle, _lib.stream_ptr()))
        return self

    def factor_shift(self) -> float:
        """Diagonal shift of the float32 factor's input: the regulariser, or 16^k times it when the float32
        factorisation of K + reg I broke down and was redone (the factor is only the preconditioner)."""
        return float(self.lib.nngp_model_factor_shift(self.handle))

    def cov_iters(self) -> int:
        """CG iterations the last predict spent continuing the covariance solve beyond the fixed sweeps (levels >= 2 do
        that when the float32 factor is a weak preconditioner; 0 = the fixed sweeps were enough)."""
        return int(self.lib.nngp_model_cov_iters(self.handle))

    def sweep_estimate(self):
        """NTK covariance: (row_rel, var_rel) -- the relative energy-norm error of the worst row and the relative variance
        error the last predict expected its fixed sweeps to leave (var_rel decides whether the rows go on by CG when the
        alpha solve took few iterations); (-1, -1) when not measured."""
        a, b = ctypes.c_double(-1.0), ctypes.c_double(-1.0)
        self._check(self.lib.nngp_model_sweep_estimate(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def fit(self, x, y):
        self.set_train(x, y)
        self.build_rows(0, self.n)
        self.factor()
        self.solve()
        return self

    # ---- checkpoint / resume (the reference keeps (X, cho_factor, alpha) only in a Python closure) ----
    def save(self, path: str):
        """Write the state that DEFINES the fit -- X, Y, architecture, ``get``, regulariser -- plus alpha as a check value.
        The factor is not written: rebuilding it (kernel build + Cholesky, 70 ms at N = 32768) is ~30x faster than reading
        the 4.3 GB float32 factor back from disk, so ``load`` recomputes it and verifies alpha against the file."""
        import torch
        if self.n == 0:
            raise _lib.NngpError("save: fit the model first")
        x, y = (t.cpu().numpy() for t in self._keep)
        w = np.array([self.arch.w_std[i] for i in range(self.arch.n_dense)])
        b = np.array([self.arch.b_std[i] for i in range(self.arch.n_dense)])
        info = self.info()
        np.savez(path, format=np.array("nngp-src_amd GPModel v1"), x=x, y=y, w_std=w, b_std=b, get=np.array(self.get),
                 diag_reg=np.array(self._diag_reg), absolute=np.array(self._absolute), n_cap=np.array(self.n_cap),
                 alpha=self.alpha().cpu().numpy(), reg=np.array(info["reg"]))
        torch.cuda.synchronize()

    @classmethod
    def load(cls, path: str, m_cap: int = 0, check: bool = True):
        """Rebuild a saved model on the current device; ``check``: alpha must agree with the saved one to 1e-8."""
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
        if str(z["format"]) != "nngp-src_amd GPModel v1":
            raise _lib.NngpError("load: %s is not a GPModel checkpoint" % path)
        x, y = z["x"], z["y"]
        model = cls(max(int(z["n_cap"]), x.shape[0]), x.shape[1], z["w_std"].tolist(), z["b_std"].tolist(), get=str(z["get"]),
                    diag_reg=float(z["diag_reg"]), diag_reg_absolute_scale=bool(z["absolute"]), ny=y.shape[1], m_cap=m_cap)
        model.fit(x, y)
        if check:
            a, a0 = model.alpha().cpu().numpy(), z["alpha"]
            err = float(np.linalg.norm(a - a0) / max(np.linalg.norm(a0), 1e-300))
            if not err < 1e-8 or abs(model.info()["reg"] - float(z["reg"])) > 1e-12 * float(z["reg"]):
                raise _lib.NngpError("load: the refitted model does not reproduce the checkpoint (alpha rel. diff %.2e)" % err)
        return model

    def kernel_buffer(self, all_rows: bool = False):
        """(torch view of the float64 train-train kernel in HBM, ld).  The view is [n, ld], or every row the
        allocation can hold at this ld when ``all_rows`` (the RCCL all-gather writes whole chunks)."""
        p, ld = ctypes.c_void_p(), ctypes.c_int64()
        self._check(self.lib.nngp_model_kernel_buffer(self.handle, ctypes.byref(p), ctypes.byref(ld)))
        np_cap = (self.n_cap + 127) // 128 * 128
        rows = (np_cap * np_cap) // ld.value if all_rows else self.n
        return _wrap_device(p.value, rows * ld.value, self.device, "<f8").view(rows, ld.value), ld.value

    def info(self) -> dict:
        fi = _lib.NngpFitInfo()
        self._check(self.lib.nngp_model_info(self.handle, ctypes.byref(fi)))
        return {k: getattr(fi, k) for k, _ in fi._fields_}

    def update_timer(self, enable=True):
        """Live HIP-event timing of the dominant kernel (the Cholesky's split-float16 trailing update); see update_timer_read."""
        self._check(self.lib.nngp_model_update_timer(self.handle, 1 if enable else 0))

    def update_timer_read(self):
        """(launches, total ms, algorithmic flops) of the trailing updates of the last factorisation (waits for it)."""
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0.0)
        fl = ctypes.c_double(0.0)
        self._check(self.lib.nngp_model_update_timer_read(self.handle, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
        return int(n.value), float(ms.value), float(fl.value)

    def update_timer_bytes(self):
        """Algorithmic bytes of those launches (C read + written once, operand split rows once)."""
        b = ctypes.c_double(0.0)
        self._check(self.lib.nngp_model_update_timer_bytes(self.handle, ctypes.byref(b)))
        return float(b.value)

    def residual_timer(self, enable=True):
        """Live HIP-event timing of the posterior's int8 plane products (k_gemm_nt_i8s); see residual_timer_read."""
        self._check(self.lib.nngp_model_residual_timer(self.handle, 1 if enable else 0))

    def residual_timer_read(self):
        """(launches, total ms, float64 flops stood for, int8 operations executed) since the last read (waits for them)."""
        n = ctypes.c_int64(0)
        ms, fl, ops = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self.lib.nngp_model_residual_timer_read(self.handle, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ops)))
        return int(n.value), float(ms.value), float(fl.value), float(ops.value)

    def trsm_timer(self, enable=True):
        """Live HIP-event timing of the posterior's blocked triangular solves; see trsm_timer_read."""
        self._check(self.lib.nngp_model_trsm_timer(self.handle, 1 if enable else 0))

    def trsm_timer_read(self):
        """(solves, total ms, algorithmic flops N^2 M each) since the last read (waits for them)."""
        n = ctypes.c_int64(0)
        ms, fl = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self.lib.nngp_model_trsm_timer_read(self.handle, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
        return int(n.value), float(ms.value), float(fl.value)

    def residual_floor(self):
        """(estimate, distrusted): what the int8 residual's dropped digit pairs may have cost this fit's level-1 variances, relative
        to them (-1.0 until a predict measured it), and whether the fit was taken off the int8 path for it."""
        r, d = ctypes.c_double(-1.0), ctypes.c_int32(0)
        self._check(self.lib.nngp_model_residual_floor(self.handle, ctypes.byref(r), ctypes.byref(d)))
        return float(r.value), bool(d.value)

    def alpha(self):
        import torch
        out = torch.empty((self.n, self.ny), dtype=torch.float64, device=self.device)
        self._check(self.lib.nngp_model_alpha(self.handle, _lib.ptr(out), _lib.stream_ptr()))
        return out

    # ---- predict_fn(x_test, get, compute_cov) ----
    def predict(self, x_test=None, cov="diag", as_numpy=True):
        """mean [M, ny] (+ var [M] for cov='diag', cov [M, M] for cov='full'/True)."""
        import torch
        mode = _COV[cov]
        if x_test is None:
            xt, m = None, self.n
        else:
            xt = _lib.to_device_f64(x_test, self.device)
            if xt.ndim != 2 or xt.shape[1] != self.d:
                raise ValueError("x_test must be [M, %d], got %s" % (self.d, tuple(xt.shape)))
            m = int(xt.shape[0])
        mean = torch.empty((m, self.ny), dtype=torch.float64, device=self.device)
        out = None
        if mode == _lib.COV_DIAG:
            out = torch.empty((m,), dtype=torch.float64, device=self.device)
        elif mode == _lib.COV_FULL:
            out = torch.empty((m, m), dtype=torch.float64, device=self.device)
        if m > 0:
            self._check(self.lib.nngp_model_predict(self.handle, _lib.ptr(xt), m, mode, _lib.ptr(mean), _lib.ptr(out),
                                                   _lib.stream_ptr()))
        if as_numpy:
            mean = mean.cpu().numpy()
            out = None if out is None else out.cpu().numpy()
        return mean if out is None else (mean, out)


    def select_pool(self, x_pool, count: int, biased: bool = False, seed: int = 10):
        """Pool scoring of the active-learning loop ON THE DEVICE (reference: active/ActiveLearner.py:43-55): predict mean and
        variance of the pool queries, score = std / max(mean), and return `count` indices -- the largest scores in ascending
        order (``np.argsort(score)[-count:]``), or, ``biased``, a score-proportional draw without replacement (Gumbel top-k on
        the counter-based generator of synth.py, seed 10 like the reference's PRNGKey(10)).  Only the indices leave the GPU."""
        import torch
        mean, var = self.predict(x_pool, cov="diag", as_numpy=False)
        m = int(mean.shape[0])
        count = min(int(count), m)
        idx = torch.empty((count,), dtype=torch.int64, device=self.device)
        if count > 0:
            self._check(self.lib.nngp_pool_select(_lib.ptr(mean), m, self.ny, _lib.ptr(var), count, int(bool(biased)), int(seed),
                                                  _lib.ptr(idx), _lib.stream_ptr()))
        return idx.cpu().numpy()


def _wrap_device(address: int, count: int, device, typestr: str):
    """Zero-copy torch view of library-owned HBM (typestr '<f8' or '<f4')."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (address, False), "version": 3,
                                  "strides": None}
    return torch.as_tensor(h, device=device)
