"""weight-gradient products (TN, long K): this library's split-K configurations against torch.matmul, us per call on an idle chip"""
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

for name, M, N, K in (("dW_hh dec", 2048, 512, 5120), ("dW_ih dec", 2048, 500, 5376), ("dW_hh enc", 1024, 256, 5120), ("dW_ih enc", 1024, 500, 5120),
                      ("attn out", 512, 1024, 5376), ("attn in", 512, 512, 5376)):
    Np = (N + 7) // 8 * 8
    A = torch.randn(K, M, device="cuda").to(T); B = torch.randn(K, Np, device="cuda").to(T)
    Cc = torch.zeros(M, N, device="cuda")
    res = []
    for tile, split in ((0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (64, 4), (64, 8), (64, 16)):
        a = L.GemmArgs()
        a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, L.GEMM_TN, A.data_ptr(), M, B.data_ptr(), Np, Cc.data_ptr(), N, M, N, K
        a.out_f32, a.alpha, a.split_k, a.tile, a.accumulate = 1, 1.0, split, tile, 1 if split == 1 else 0
        res.append("%s/%d: %5.1f" % ("t64" if tile else "auto", split, timeit(lambda: L.check(lib.vmmt_gemm(C.byref(a), None), "g"))))
    At = A.t()
    t2 = timeit(lambda: torch.matmul(At, B))
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    eng = max(1, min(K // 256, (1024 + tiles - 1) // tiles))
    print("%-10s %5d x %4d x %5d  engine split %2d | %s | torch %5.1f" % (name, M, N, K, eng, "  ".join(res), t2))
