"""the dWg product's shape [30000 x 512 x 5120] (bf16, f32 out, 256 x 128 tiles) in the three operand layouts: what would P stored as P^T buy?
python tools/dwg_layouts.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
M, N, K = 30000, 512, 5120
T = torch.bfloat16
torch.manual_seed(0)
ldm = 30016
bufs = {
    "TN": (torch.randn(K, ldm, device="cuda").to(T), ldm, torch.randn(K, N, device="cuda").to(T), N, L.GEMM_TN),
    "NN": (torch.randn(M, K, device="cuda").to(T), K, torch.randn(K, N, device="cuda").to(T), N, L.GEMM_NN),
    "NT": (torch.randn(M, K, device="cuda").to(T), K, torch.randn(N, K, device="cuda").to(T), K, L.GEMM_NT),
}
Cc = torch.zeros(M, N, device="cuda")
TILES = [int(x) for x in os.environ.get("TILES", "256,128").split(",")]      # 520 (TN only): the probe build's one-wave-per-SIMD pipeline for K-strided operands
ref = None
for tile in TILES:
    for name, (A, lda, B, ldb, lay) in bufs.items():
        if tile == 520 and name != "TN":
            continue
        a = L.GemmArgs()
        a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, lay, A.data_ptr(), lda, B.data_ptr(), ldb, Cc.data_ptr(), N, M, N, K
        a.out_f32, a.alpha, a.tile = 1, 1.0, tile
        for _ in range(3):
            L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        chk = ""
        if name == "TN":                       # the layouts share no operands: only TN is checked against torch (bf16 products, f32 sums)
            if ref is None:
                ref = torch.matmul(A[:, :M].t().float(), B.float())
            err = float((Cc - ref).abs().max() / ref.abs().max())
            assert err < 2e-2, (tile, err)
            chk = "  (max err %.1e of max)" % err
        print("tile %d %s: %.1f us  %.0f TFLOP/s%s" % (tile, name, us, 2.0 * M * N * K / us / 1e6, chk))
