"""the step's weight-gradient products (TN: dW[M x N] += dgates[K x M]^T x[K x N], K = tokens of the batch) in isolation, us per call by
split-K depth, against torch.matmul (hipBLASLt) on the same operands: python tools/wgrad_gemm.py"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
T = torch.bfloat16
torch.manual_seed(0)


def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


SHAPES = (("dW_hh dec", 2048, 512, 5120), ("dW_ih dec", 2048, 768, 5120), ("dW_ih enc (dir)", 1024, 512, 5120), ("dW_hh enc (dir)", 1024, 256, 5120),
          ("dW_out", 512, 1024, 5120), ("dW_a", 512, 512, 5120), ("dW_ih dec cfg5", 4096, 1536, 16384), ("dW_hh dec cfg5", 4096, 1024, 16384))
print("xsplit env:", os.environ.get("VMMT_GEMM_XSPLIT", "0"))
for name, M, N, K in SHAPES:
    A = torch.randn(K, M, device="cuda").to(T); B = torch.randn(K, N, device="cuda").to(T)
    ref = (A.float().t() @ B.float())
    res = []
    for sk in (1, 2, 4, 8, 16):
        Cc = torch.zeros(M, N, device="cuda")
        a = L.GemmArgs()
        a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, L.GEMM_TN, A.data_ptr(), M, B.data_ptr(), N, Cc.data_ptr(), N, M, N, K
        a.out_f32, a.alpha, a.split_k, a.accumulate = 1, 1.0, sk, 1
        f = lambda: L.check(lib.vmmt_gemm(C.byref(a), None), "g")
        f(); torch.cuda.synchronize()
        err = ((Cc - ref).abs().max() / ref.abs().max()).item()
        res.append("k%d %6.1f%s" % (sk, timeit(f), "" if err < 2e-2 else " WRONG(%.2g)" % err))
    At = A.t()
    t2 = timeit(lambda: torch.matmul(At, B))
    print("%-18s %5d x %5d x %6d  %s | torch %6.1f us" % (name, M, N, K, "  ".join(res), t2), flush=True)
