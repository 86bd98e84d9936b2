"""one GEMM shape, many launches -- for PMC runs"""
import ctypes as C, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "dWg"
T = torch.bfloat16
if which == "dWg":
    M, N, K, lay = 30000, 512, 5120, L.GEMM_NN
    A = torch.randn(M, K, device='cuda').to(T); B = torch.randn(K, N, device='cuda').to(T); lda, ldb = K, N
else:
    M, N, K, lay = 30000, 5120, 512, L.GEMM_NT
    A = torch.randn(M, K, device='cuda').to(T); B = torch.randn(N, K, device='cuda').to(T); lda, ldb = K, K
Cc = torch.zeros(M, N, device='cuda', dtype=torch.float32 if which == "dWg" else T)
a = L.GemmArgs(L.BF16, lay, A.data_ptr(), lda, B.data_ptr(), ldb, Cc.data_ptr(), N, M, N, K, 0, 0, None, 0, 0, 0, 0, 1 if which == "dWg" else 0, 0, 1.0, None, 1, int(sys.argv[2]) if len(sys.argv) > 2 else 0, 0)
for _ in range(6): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
torch.cuda.synchronize()
