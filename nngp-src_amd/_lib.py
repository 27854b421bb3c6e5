"""ctypes binding of ``libnngp_hip.so`` (C ABI in ``include/nngp_hip.h``).

There is no CPU fallback: every compute entry point raises if the HIP library is missing or no
MI355X is visible.  PyTorch is used only as the owner of device memory and streams.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnngp_hip.so")
KNOBS_LIB_PATH = os.path.join(_HERE, "libnngp_hip_knobs.so")  # same sources + timing knobs; scripts/ and A/B tests only

GET_NNGP, GET_NTK = 1, 2
DTYPE_F32, DTYPE_F64 = 0, 1
COV_NONE, COV_DIAG, COV_FULL = 0, 1, 2
MAX_DENSE = 16

# every symbol include/nngp_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "nngp_version", "nngp_last_error", "nngp_kernel_build", "nngp_kernel_diag", "nngp_model_create",
    "nngp_model_destroy", "nngp_model_fit", "nngp_model_set_train", "nngp_model_build_rows",
    "nngp_model_factor", "nngp_model_factor_begin", "nngp_model_factor_panel", "nngp_model_factor_update",
    "nngp_model_factor_end", "nngp_model_factor_buffers", "nngp_model_solve", "nngp_model_append", "nngp_model_kernel_buffer", "nngp_model_info",
    "nngp_model_alpha", "nngp_model_predict", "nngp_model_set_refine", "nngp_model_cov_iters", "nngp_model_sweep_estimate", "nngp_model_factor_shift", "nngp_model_prepare_serving", "nngp_potrf_f32", "nngp_gemm_nt_f32",
    "nngp_gemm_nt_h3", "nngp_gemm_nt_f64", "nngp_trsm_rlt_f32", "nngp_model_apply_factor",
    "nngp_model_factor_input_rows", "nngp_model_factor_input_complete", "nngp_model_precond", "nngp_model_matvec_rows", "nngp_model_set_alpha", "nngp_encoder_create", "nngp_encoder_destroy", "nngp_encoder_dim",
    "nngp_encoder_encode", "nngp_comm_unique_id", "nngp_comm_create", "nngp_comm_destroy", "nngp_comm_library",
    "nngp_allgather_rows", "nngp_bcast", "nngp_model_update_timer", "nngp_model_update_timer_read", "nngp_symv_f64",
    "nngp_pool_select", "nngp_model_update_timer_bytes", "nngp_model_factor_update_cols", "nngp_gemm_nt_i8s", "nngp_model_residual_timer", "nngp_model_residual_timer_read", "nngp_model_trsm_timer", "nngp_model_trsm_timer_read", "nngp_model_residual_floor",
    "nngp_trsm_ticket_order", "nngp_trsm_ticket_queues", "nngp_model_reserve", "nngp_alloc_count",
)


class NngpArch(ctypes.Structure):
    _fields_ = [("n_dense", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("w_std", ctypes.c_double * MAX_DENSE), ("b_std", ctypes.c_double * MAX_DENSE)]


class NngpFitInfo(ctypes.Structure):
    _fields_ = [("reg", ctypes.c_double), ("trace_mean", ctypes.c_double), ("rel_residual", ctypes.c_double),
                ("refine_iters", ctypes.c_int32), ("clamped_pivots", ctypes.c_int32),
                ("n", ctypes.c_int64), ("n_padded", ctypes.c_int64)]


class NngpError(RuntimeError):
    pass


_libs = {}


def load(knobs: bool = False):
    """Load libnngp_hip.so (no GPU needed to load; compute calls need one).

    ``knobs=True`` loads libnngp_hip_knobs.so instead -- the same sources built with -DNNGP_TIMING_KNOBS, which adds
    ``nngp_debug_set`` (ablations for scripts/ and the A/B tests; some produce wrong results on purpose).  The product
    library has no such entry point.  NNGP_KNOBS=1 in the environment makes it the default (scripts/ only)."""
    knobs = bool(knobs) or os.environ.get("NNGP_KNOBS", "0") == "1"
    if knobs in _libs:
        return _libs[knobs]
    path = KNOBS_LIB_PATH if knobs else LIB_PATH
    if not os.path.exists(path):
        raise NngpError(
            "%s is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nngp-src_amd/csrc`; there is no CPU fallback." % (os.path.basename(path), path))
    # torch first: its wheel carries its own libamdhip64 / libhsa-runtime64.  Loaded in that order the library's
    # DT_NEEDED entries resolve to the copies torch already mapped (same SONAME); the other way round the process ends
    # up with two HSA runtimes and the second one reports "no ROCm-capable device" (seen on the GPU box).
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    bind_prototypes(lib, knobs)
    _libs[knobs] = lib
    return lib


def bind_prototypes(lib, knobs: bool = False):
    """Argument and result types of every entry point of include/nngp_hip.h on a loaded library.  Also applied by the test
    infrastructure to its host build of the same ABI, so that both sit behind one interface."""
    vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
    archp = ctypes.POINTER(NngpArch)
    lib.nngp_version.restype = ctypes.c_int
    lib.nngp_last_error.restype = ctypes.c_char_p
    if knobs:
        lib.nngp_debug_set.argtypes = [i32, i32]
        lib.nngp_debug_set.restype = ctypes.c_int
    lib.nngp_kernel_build.argtypes = [vp, i64, vp, i64, i32, archp, i32, vp, vp, i64, i64, i64, vp]
    lib.nngp_kernel_diag.argtypes = [vp, i64, i32, archp, vp, vp, vp]
    lib.nngp_model_create.argtypes = [ctypes.POINTER(vp), i64, i64, i32, i32, archp, i32, dbl, i32]
    lib.nngp_model_destroy.argtypes = [vp]
    lib.nngp_model_fit.argtypes = [vp, vp, vp, i64, vp]
    lib.nngp_model_set_train.argtypes = [vp, vp, vp, i64, vp]
    lib.nngp_model_build_rows.argtypes = [vp, i64, i64, vp]
    lib.nngp_model_factor.argtypes = [vp, vp]
    lib.nngp_model_factor_begin.argtypes = [vp, vp]
    lib.nngp_model_factor_panel.argtypes = [vp, i64, i64, vp]
    lib.nngp_model_factor_update.argtypes = [vp, i64, i64, i64, i64, vp]
    lib.nngp_model_factor_end.argtypes = [vp, vp]
    lib.nngp_model_factor_buffers.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(vp)]
    lib.nngp_model_solve.argtypes = [vp, i32, dbl, vp]
    lib.nngp_model_kernel_buffer.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.nngp_model_info.argtypes = [vp, ctypes.POINTER(NngpFitInfo)]
    lib.nngp_model_alpha.argtypes = [vp, vp, vp]
    lib.nngp_model_predict.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.nngp_encoder_create.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, i32, i32]
    lib.nngp_encoder_destroy.argtypes = [vp]
    lib.nngp_encoder_dim.argtypes = [vp]
    lib.nngp_encoder_encode.argtypes = [vp, ctypes.c_char_p, i64, i32, vp, vp, i64, ctypes.POINTER(i64)]
    lib.nngp_model_set_refine.argtypes = [vp, i32]
    lib.nngp_model_cov_iters.argtypes = [vp]
    lib.nngp_model_prepare_serving.argtypes = [vp, vp]
    lib.nngp_model_factor_shift.argtypes = [vp]
    lib.nngp_model_factor_shift.restype = ctypes.c_double
    lib.nngp_model_cov_iters.restype = ctypes.c_int
    lib.nngp_model_sweep_estimate.argtypes = [vp, ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    lib.nngp_model_append.argtypes = [vp, vp, vp, i64, vp]
    lib.nngp_gemm_nt_f64.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, dbl, dbl, vp]
    lib.nngp_gemm_nt_i8s.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, dbl, dbl, i32, i32, i32, vp]
    lib.nngp_potrf_f32.argtypes = [vp, i64, i64, vp, vp, vp]
    lib.nngp_gemm_nt_f32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, ctypes.c_float, ctypes.c_float, i32, vp]
    lib.nngp_gemm_nt_h3.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, i32, vp]
    lib.nngp_trsm_rlt_f32.argtypes = [vp, i64, i64, vp, i64, vp, i64, vp]
    lib.nngp_model_apply_factor.argtypes = [vp, vp, i64, i32, vp]
    lib.nngp_model_factor_input_rows.argtypes = [vp, i64, i64, ctypes.c_double, vp]
    lib.nngp_model_factor_input_complete.argtypes = [vp]
    lib.nngp_model_precond.argtypes = [vp, vp, vp, vp]
    lib.nngp_model_matvec_rows.argtypes = [vp, vp, vp, i64, i64, vp]
    lib.nngp_model_set_alpha.argtypes = [vp, vp, i32, ctypes.c_double, vp]
    lib.nngp_comm_unique_id.argtypes = [vp]
    lib.nngp_comm_create.argtypes = [ctypes.POINTER(vp), vp, i32, i32]
    lib.nngp_comm_destroy.argtypes = [vp]
    lib.nngp_comm_library.restype = ctypes.c_char_p
    lib.nngp_allgather_rows.argtypes = [vp, i64, i64, i32, vp, vp]
    lib.nngp_bcast.argtypes = [vp, i64, i32, i32, vp, vp]
    lib.nngp_symv_f64.argtypes = [vp, i64, i64, vp, vp, dbl, vp]
    lib.nngp_pool_select.argtypes = [vp, i64, i32, vp, i64, i32, ctypes.c_uint64, vp, vp]
    lib.nngp_model_update_timer.argtypes = [vp, i32]
    lib.nngp_model_update_timer_read.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    lib.nngp_model_update_timer_bytes.argtypes = [vp, ctypes.POINTER(dbl)]
    lib.nngp_model_residual_timer.argtypes = [vp, i32]
    lib.nngp_model_residual_floor.argtypes = [vp, ctypes.POINTER(dbl), ctypes.POINTER(i32)]
    lib.nngp_model_residual_timer_read.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(dbl), ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    lib.nngp_model_trsm_timer.argtypes = [vp, i32]
    lib.nngp_model_trsm_timer_read.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    lib.nngp_model_reserve.argtypes = [vp, i64, i32]
    lib.nngp_alloc_count.argtypes = []
    lib.nngp_trsm_ticket_order.argtypes = [i32, i32, i32, i32, i32, i32, vp, i64, ctypes.POINTER(i64)]
    lib.nngp_trsm_ticket_queues.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, vp, i64, ctypes.POINTER(i64)]
    lib.nngp_model_factor_update_cols.argtypes = [vp, i64, i64, ctypes.POINTER(i64), i32, i64, vp]
    for name in ABI_SYMBOLS:
        if name not in ("nngp_last_error", "nngp_model_factor_shift", "nngp_comm_library", "nngp_alloc_count"):
            getattr(lib, name).restype = ctypes.c_int
    lib.nngp_alloc_count.restype = ctypes.c_int64
    return lib


def check(rc: int, lib=None):
    if rc != 0:
        msg = (lib or load()).nngp_last_error()
        raise NngpError("libnngp_hip: rc=%d: %s" % (rc, (msg or b"").decode("utf-8", "replace")))


def make_arch(w_std, b_std) -> NngpArch:
    w_std = [float(v) for v in w_std]
    b_std = [float(v) for v in b_std]
    if len(w_std) != len(b_std) or not 1 <= len(w_std) <= MAX_DENSE:
        raise ValueError("architecture must have 1..%d Dense layers" % MAX_DENSE)
    arch = NngpArch()
    arch.n_dense = len(w_std)
    for i, (w, b) in enumerate(zip(w_std, b_std)):
        arch.w_std[i] = w
        arch.b_std[i] = b
    return arch


def require_gpu():
    """The torch device used for HBM buffers; raises (never falls back) when no GPU is present."""
    import torch
    if not torch.cuda.is_available():
        raise NngpError("no MI355X/ROCm device visible: the NNGP hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_f64(a, device=None):
    """numpy / torch array -> contiguous float64 device tensor."""
    import torch
    device = device or require_gpu()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())
