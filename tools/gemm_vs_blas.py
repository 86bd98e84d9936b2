"""the step's large plain GEMMs: this library's kernels against torch.matmul (hipBLASLt / rocBLAS) on the same operands, us per call"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
T = torch.bfloat16
torch.manual_seed(0)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def ours(lay, A, lda, B, ldb, M, N, K, out_f32=1, split_k=0):
    Cc = torch.zeros(M, N, device="cuda", dtype=torch.float32 if out_f32 else T)
    a = L.GemmArgs()
    a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, lay, A.data_ptr(), lda, B.data_ptr(), ldb, Cc.data_ptr(), N, M, N, K
    a.out_f32, a.alpha, a.split_k = out_f32, 1.0, split_k
    return (lambda: L.check(lib.vmmt_gemm(C.byref(a), None), "g")), Cc

rows = []
# TN: C[M x N] = A^T B, A stored [K x M], B stored [K x N]
for name, M, N, K in (("dWg  P^T O", 30000, 512, 5376), ("dW_hh dec", 2048, 512, 5120), ("dW_ih dec", 2048, 500, 5376), ("dW_ih enc", 2048, 500, 5120)):
    Np = (N + 7) // 8 * 8
    A = torch.randn(K, M, device="cuda").to(T); B = torch.randn(K, Np, device="cuda").to(T)
    f, Cc = ours(L.GEMM_TN, A, M, B, Np, M, N, K, 1, -1 if M < 30000 else 0)
    t1 = timeit(f)
    At = A.t()
    t2 = timeit(lambda: torch.matmul(At, B))
    rows.append((name + " (TN)", M, N, K, t1, t2))
# NT: C = A B^T, A [M x K], B [N x K]
for name, M, N, K in (("gx enc", 5120, 2048, 512), ("dO = G Wg", 5376, 512, 30016)):
    A = torch.randn(M, K, device="cuda").to(T); B = torch.randn(N, K, device="cuda").to(T)
    f, Cc = ours(L.GEMM_NT, A, K, B, K, M, N, K, 0)
    t1 = timeit(f)
    Bt = B.t()
    t2 = timeit(lambda: torch.matmul(A, Bt))
    rows.append((name + " (NT)", M, N, K, t1, t2))
# NN: C = A B, A [M x K], B [K x N]
for name, M, N, K in (("dX = dg W", 5120, 512, 2048),):
    A = torch.randn(M, K, device="cuda").to(T); B = torch.randn(K, N, device="cuda").to(T)
    f, Cc = ours(L.GEMM_NN, A, K, B, N, M, N, K, 0)
    t1 = timeit(f)
    t2 = timeit(lambda: torch.matmul(A, B))
    rows.append((name + " (NN)", M, N, K, t1, t2))
for name, M, N, K, t1, t2 in rows:
    fl = 2.0 * M * N * K
    print("%-18s %6d x %5d x %6d   ours %7.1f us (%5.0f TF/s)   torch %7.1f us (%5.0f TF/s)" % (name, M, N, K, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6))
