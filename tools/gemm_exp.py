"""times one GEMM shape/tile (results are garbage for the experiment builds)"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
T = torch.bfloat16
M, N, K, lay = 30000, 512, 5120, L.GEMM_NN
A = torch.randn(M, K, device='cuda').to(T); B = torch.randn(K, N, device='cuda').to(T)
Cc = torch.zeros(M, N, device='cuda', dtype=torch.float32)
for tile in [int(x) for x in sys.argv[1:]]:
    a = L.GemmArgs(L.BF16, lay, A.data_ptr(), K, B.data_ptr(), N, Cc.data_ptr(), N, M, N, K, 0, 0, None, 0, 0, 0, 0, 1, 0, 1.0, None, 1, tile, 0)
    for _ in range(3): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
    e.record(); torch.cuda.synchronize()
    print("%s tile %d: %.1f us" % (os.environ.get("VMMT_LIB_PATH", "base")[-14:], tile, s.elapsed_time(e) * 100))
