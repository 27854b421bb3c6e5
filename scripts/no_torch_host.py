#!/usr/bin/env python3
"""A host without PyTorch: bind libnngp_hip.so with ctypes alone (system HIP runtime), create a model, read its info.
Checks that the C ABI does not depend on torch being in the process."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "nngp-src_amd", "libnngp_hip.so"))
assert "torch" not in sys.modules


class NngpArch(ctypes.Structure):
    _fields_ = [("n_dense", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("w_std", ctypes.c_double * 16), ("b_std", ctypes.c_double * 16)]


arch = NngpArch(n_dense=2)
arch.w_std[0] = arch.w_std[1] = 1.0
vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
lib.nngp_model_create.argtypes = [ctypes.POINTER(vp), i64, i64, i32, i32, ctypes.POINTER(NngpArch), i32, dbl, i32]
lib.nngp_model_destroy.argtypes = [vp]
lib.nngp_last_error.restype = ctypes.c_char_p
h = vp()
rc = lib.nngp_model_create(ctypes.byref(h), 1000, 200, 20, 1, ctypes.byref(arch), 1, 1e-3, 0)
print("nngp_model_create rc=%d %s" % (rc, (lib.nngp_last_error() or b"").decode()))
if rc == 0:
    lib.nngp_model_destroy(h)
sys.exit(0 if rc == 0 else 1)
