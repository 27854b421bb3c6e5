#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel for one bench step.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE reports half of the bytes
of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, cfg):
    files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            val = float(row.get("Counter_Value", 0) or 0)
            agg[name][0] += 1
            agg[name][1] += val
    return agg


def short(name):
    if "k_build_mfma" in name:
        return "k_build_mfma"
    for key in ("k_gemm_nt_f32<2, 2, false>", "k_gemm_nt_f32<2, 2, true>", "k_gemm_nt_f32<1, 2, false>", "k_gemm_nt_f32<1, 1, false>",
                "k_gemm_nt_f32<1, 1, true>", "k_gemm_nt_f64", "k_potrf_leaf", "k_build", "k_gemv_f64", "k_gemv_n_f32",
                "k_gemv_t_partial_f32", "k_factor_input", "k_transpose_f32", "k_gemm_nt_h3v2<true", "k_gemm_nt_h3v2<false", "k_gemm_nt_h3<true>", "k_gemm_nt_h3<false>", "k_split_rows",
                "k_trsm_panel_f32", "k_trsm_panel_h3", "k_split_diag_frag", "k_symv_tiles_f64", "k_symv_reduce_f64", "k_split_lower_t",
                "k_gemm_nt_i8s", "k_i8s_slice_rows", "k_i8s_slice_sym", "k_i8s_combine", "k_trsm_tickets"):
        if key in name:
            return {"k_gemm_nt_h3v2<true": "k_gemm_nt_h3v2<true>", "k_gemm_nt_h3v2<false": "k_gemm_nt_h3v2<false>"}.get(key, key)
    return None


def main():
    fetch, write, cfg = load(sys.argv[1], sys.argv[3]), load(sys.argv[2], sys.argv[3]), sys.argv[3]
    out = {"config": cfg, "unit": "bytes per bench step (1 step, no warm-up)", "fetch_correction": "x2 (gfx950 wide reads)", "kernels": {}}
    for name, (calls, kib) in fetch.items():
        k = short(name)
        if k is None:
            continue
        e = out["kernels"].setdefault(k, {"calls": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
        e["calls"] += calls
        e["fetch_bytes"] += kib * 1024 * 2
    for name, (calls, kib) in write.items():
        k = short(name)
        if k is None:
            continue
        e = out["kernels"].setdefault(k, {"calls": calls, "fetch_bytes": 0.0, "write_bytes": 0.0})
        e["write_bytes"] += kib * 1024
    chol = [k for k in out["kernels"] if k.startswith("k_gemm_nt_f32") or k in ("k_potrf_leaf", "k_gemm_nt_h3<true>", "k_gemm_nt_h3v2<true>", "k_split_rows", "k_trsm_panel_f32", "k_trsm_panel_h3", "k_split_diag_frag")]
    out["cholesky_bytes_note"] = ("trailing updates (k_gemm_nt_h3<true>) + fused panel solves + float32 GEMMs + leaf + split kernels of the step "
                                  "(the posterior's few float32 GEMMs included; its split-float16 solves, k_gemm_nt_h3<false>, are not). "
                                  "FETCH_SIZE counts what leaves L2 toward the fabric, Infinity-Cache hits included (MI355X_MICROARCH.md), so this is "
                                  "an upper bound on HBM traffic; the x2 applies to 16-byte-per-lane reads, which all of these kernels use")
    out["cholesky_bytes"] = sum(out["kernels"][k]["fetch_bytes"] + out["kernels"][k]["write_bytes"] for k in chol)
    # fingerprint of the kernel sources these counters were taken with: bench.py only quotes the traffic while they are unchanged
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nngp-src_amd", "csrc")
    out["sources_sha16"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16]
                            for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
