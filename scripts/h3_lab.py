"""Round-3 lab for the split-float16 GEMM forms (debug key 5 = 41: k_gemm_nt_h3, 42: k_gemm_nt_h3v2).
  check : both forms against a float64 product on edge shapes (exact on small integers, lower-only masks, A = I with asymmetric B)
  time  : interleaved rounds of the configurations in H3_CONFIGS ("5=41;5=42;5=42,0=8" ...), to be run under
          rocprofv3 --kernel-trace; scripts/h3_lab_report.py matches the trace to the plan by dispatch order."""
import os; os.environ.setdefault("NNGP_KNOBS", "1")
import json, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from nngp_src_amd import _lib
lib = _lib.load(knobs=True)
dev = G.dev()


def set_cfg(cfg):
    for k in range(8):
        _lib.check(lib.nngp_debug_set(k, 0))
    for kv in cfg.split(","):
        if kv:
            k, v = kv.split("=")
            _lib.check(lib.nngp_debug_set(int(k), int(v)))


def h3(c, a, b, alpha, beta, scale, lower=False):
    m, k = a.shape
    _lib.check(lib.nngp_gemm_nt_h3(_lib.ptr(c), c.stride(0), _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), m, b.shape[0], k,
                                   alpha, beta, scale, int(lower), _lib.stream_ptr()))
    torch.cuda.synchronize()


def check():
    torch.manual_seed(11)
    worst = {}
    for form in (41, 42):
        set_cfg("5=%d" % form)
        # exact integer data, edge tiles, A = I with asymmetric B
        m, n, k = 384, 640, 96
        b = torch.randint(-8, 9, (n, k), device=dev).float()
        a = torch.zeros((m, k), device=dev); a[torch.arange(96), torch.arange(96)] = 1.0
        c = torch.full((m, n), float("nan"), device=dev)
        h3(c, a, b, 1.0, 0.0, 1.0)
        assert torch.equal(c[:96], b[:, :96].T.contiguous()) and torch.all(c[96:] == 0), form
        a = torch.randint(-8, 9, (m, k), device=dev).float()
        c0 = torch.randint(-8, 9, (m, n), device=dev).float(); c = c0.clone()
        h3(c, a, b, -1.0, 1.0, 4.0)
        assert torch.equal(c.double(), c0.double() - a.double() @ b.double().T), form
        for (m, n, k) in [(128, 128, 32), (1024, 768, 512), (2304, 2304, 1024), (2560, 2560, 64), (4096 + 128, 4096 + 128, 1024)]:
            sr = torch.exp2(torch.randint(-6, 1, (m, 1), device=dev).float())
            a = torch.randn((m, k), device=dev) * sr; b = torch.randn((n, k), device=dev)
            c0 = torch.randn((m, n), device=dev); c = c0.clone()
            h3(c, a, b, -1.0, 1.0, 2.0 ** 10)
            ref = c0.double() - a.double() @ b.double().T
            bound = c0.abs().double() + a.abs().double() @ b.abs().double().T
            tol = 2.0 ** -20 + 4 * (k + 1) ** 0.5 * 6e-8
            err = ((c.double() - ref).abs() / bound).max().item()
            assert err <= tol, (form, m, n, k, err, tol)
            worst[(form, m, n, k)] = err
            if m == n:
                c = torch.zeros((m, m), device=dev)
                h3(c, a, a, 1.0, 0.0, 2.0 ** 10, lower=True)
                ref = a.double() @ a.double().T
                idx = torch.arange(m, device=dev)
                el = idx[:, None] >= idx[None, :]
                tu = (idx[:, None] // 128) < (idx[None, :] // 128)
                bound = a.abs().double() @ a.abs().double().T
                assert ((c.double() - ref).abs() <= tol * bound)[el].all(), (form, m, "lower")
                assert torch.all(c[tu] == 0), (form, m, "mask")
                del ref, el, tu, bound
            del a, b, c, c0
    # the two forms against each other on one large lower update, repeated (race screen: results must not change run to run)
    m, k = 8192, 1024
    a = torch.randn((m, k), device=dev)
    outs = []
    for form in (41, 42, 42, 42, 42):
        set_cfg("5=%d" % form)
        c = torch.zeros((m, m), device=dev)
        h3(c, a, a, -1.0, 0.0, 2.0 ** 10, lower=True)
        outs.append(c)
    for o in outs[2:]:
        assert torch.equal(o, outs[1]), "form 2 is not reproducible"
    d = (outs[0].double() - outs[1].double()).abs().max().item() / outs[0].abs().max().item()
    print("check ok; worst rel errors:", {str(k_): "%.2e" % v for k_, v in worst.items()}, "forms differ by %.2e" % d)
    set_cfg("")


def time_plan():
    cfgs = os.environ.get("H3_CONFIGS", "5=41;5=42").split(";")
    shapes = [(30720, 30720, 1024, True), (16384, 16384, 1024, True), (1024, 31744, 1024, False)]
    if os.environ.get("H3_SHAPES"):  # "m,n,k,lower;..."
        shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["H3_SHAPES"].split(";")]
        shapes = [(m, n, k, bool(lo)) for (m, n, k, lo) in shapes]
    rounds = int(os.environ.get("H3_ROUNDS", "5"))
    plan = []
    for (m, n, k, lower) in shapes:
        a = torch.randn((m, k), device=dev)
        b = a if lower else torch.randn((n, k), device=dev)
        c = torch.zeros((m, n), device=dev)
        for rnd in range(rounds):
            for cfg in cfgs:
                set_cfg(cfg)
                _lib.check(lib.nngp_gemm_nt_h3(_lib.ptr(c), c.stride(0), _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), m, n, k,
                                               -1.0, 1.0, 2.0 ** 10, int(lower), _lib.stream_ptr()))
                torch.cuda.synchronize()
                plan.append({"shape": [m, n, k, lower], "cfg": cfg, "round": rnd})
        del a, b, c
    set_cfg("")
    json.dump(plan, open(os.path.join(ROOT, "gpurun_out", "h3_lab_plan.json"), "w"))


if __name__ == "__main__":
    if sys.argv[1] == "check":
        check()
    else:
        time_plan()
