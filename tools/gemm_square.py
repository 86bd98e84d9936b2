"""where this library's bf16 GEMM loop stands on shapes that are NOT the step's: large squares and the dWg shape, NT layout, against torch.matmul
(hipBLASLt).  python tools/gemm_square.py"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
T = torch.bfloat16
torch.manual_seed(0)

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (30000, 512, 5120), (30000, 2048, 5120), (8192, 512, 8192)):
    A = torch.randn(M, K, device="cuda").to(T); B = torch.randn(N, K, device="cuda").to(T)
    if os.environ.get("ZEROS") == "1":       # all-zero operands: the same instruction stream at a fraction of the switching power
        A.zero_(); B.zero_()
    Bt = B.t()
    res = []
    tiles = [int(x) for x in os.environ.get("TILES", "256,128").split(",")]       # 512 / 513: the probe build (tools/exp_build.sh gemm.hip TILE512)
    ref = torch.matmul(A, Bt).float()
    for tile in tiles:
        Cc = torch.zeros(M, N, device="cuda", dtype=T)
        a = L.GemmArgs()
        a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, L.GEMM_NT, A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, M, N, K
        a.out_f32, a.alpha, a.tile = 0, 1.0, tile
        res.append(timeit(lambda: L.check(lib.vmmt_gemm(C.byref(a), None), "g")))
        err = float((Cc.float() - ref).abs().max() / max(1e-30, float(ref.abs().max())))
        assert err < 2e-2 or os.environ.get("NOCHECK") == "1", (tile, err)
    t2 = timeit(lambda: torch.matmul(A, Bt))
    fl = 2.0 * M * N * K / 1e6
    print("%6d x %5d x %5d  " % (M, N, K) + " | ".join("tile %d: %7.1f us %5.0f TF" % (t, r, fl / r) for t, r in zip(tiles, res)) +
          " | torch.matmul %7.1f us %5.0f TF" % (t2, fl / t2))
