"""What a dense float16 GEMM reaches on this box through the vendor library (torch.matmul -> hipBLASLt), next to the
nominal 2.5 PFLOP/s the roofline object prices against: calibration for the k_gemm_nt_h3 figure (three float16 MFMA
products per output tile, fp32 accumulate).  Prints TFLOP/s for square and for trailing-update-like shapes."""
import json, torch
dev = torch.device("cuda", 0)
def bench(m, n, k, dtype, reps=20):
    a = torch.randn(m, k, device=dev, dtype=dtype); b = torch.randn(n, k, device=dev, dtype=dtype)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(m=m, n=n, k=k, dtype=str(dtype).split(".")[-1], ms=round(ms, 4), tflops=round(2.0 * m * n * k / ms / 1e9, 1))
out = []
for dtype in (torch.float16, torch.bfloat16):
    for (m, n, k) in ((8192, 8192, 8192), (16384, 16384, 8192), (16384, 16384, 1024), (28672, 28672, 1024), (16384, 16384, 3072), (28672, 28672, 3072)):
        out.append(bench(m, n, k, dtype)); print(json.dumps(out[-1]), flush=True)
