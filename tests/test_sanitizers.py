"""CPU-only: the host C++ of the product (csrc/encoder.cpp) and the C oracle (oracle/nngp_oracle.c) built with
-fsanitize=address,undefined and driven through their C entry points -- golden vectors, numpy-oracle agreement, malformed
and mutated input.  (GPU AddressSanitizer is not available on this pool; device code is covered by the parity tests.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def _libpath(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
    return p if os.path.isabs(p) else None


@pytest.mark.skipif(_libpath("libasan.so") is None or _libpath("libubsan.so") is None, reason="no sanitizer runtime")
def test_host_cxx_and_c_oracle_under_asan_ubsan(tmp_path):
    host_so, oracle_so = str(tmp_path / "libhost_san.so"), str(tmp_path / "liboracle_san.so")
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-Wall"] + SAN + [
        os.path.join(ROOT, "nngp-src_amd", "csrc", "encoder.cpp"), os.path.join(ROOT, "tests", "sanitize", "shim.cpp"), "-o", host_so])
    subprocess.check_call(["gcc", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-Wall", "-Wextra"] + SAN + [
        os.path.join(ROOT, "oracle", "nngp_oracle.c"), "-o", oracle_so, "-lm"])
    env = dict(os.environ)
    env["LD_PRELOAD"] = _libpath("libasan.so") + ":" + _libpath("libubsan.so")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=66"  # the interpreter itself leaks by design
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:exitcode=67"
    env["OMP_NUM_THREADS"] = "4"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "worker.py"), host_so, oracle_so],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SANITIZE_WORKER_OK" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
