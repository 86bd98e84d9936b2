"""a large TN (weight-gradient) product on the 128 x 128 two-stage and the 256 x 128 three-stage tiles, by split-K:
python tools/gemm_tn_tiles.py [M N K]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (4096, 1024, 16384)
T = torch.bfloat16
torch.manual_seed(0)
A = torch.randn(K, M, device="cuda").to(T); B = torch.randn(K, N, device="cuda").to(T)
Cc = torch.zeros(M, N, device="cuda")
for tile in (0, 128, 256):
    for split in (1, 2, 3, 4):
        a = L.GemmArgs()
        a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, L.GEMM_TN, A.data_ptr(), M, B.data_ptr(), N, Cc.data_ptr(), N, M, N, K
        a.out_f32, a.alpha, a.tile, a.split_k, a.accumulate = 1, 1.0, tile, split, 1 if split == 1 else 0
        for _ in range(3):
            L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(10):
            L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        print("M %d N %d K %d tile %3d split %d: %7.1f us  %6.1f TFLOP/s" % (M, N, K, tile, split, us, 2.0 * M * N * K / us / 1e6), flush=True)
