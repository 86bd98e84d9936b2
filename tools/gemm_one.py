"""one large TN product, launched a few times (for rocprofv3 --pmc passes: tools/pmc_gemm.sh): python tools/gemm_one.py [M N K]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (30000, 512, 5376)
T = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(K, M, device="cuda").to(T); B = torch.randn(K, N, device="cuda").to(T)
Cc = torch.zeros(M, N, device="cuda")
a = L.GemmArgs()
a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, L.GEMM_TN, A.data_ptr(), M, B.data_ptr(), N, Cc.data_ptr(), N, M, N, K
a.out_f32, a.alpha = 1, 1.0
for _ in range(8):
    L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
torch.cuda.synchronize()
