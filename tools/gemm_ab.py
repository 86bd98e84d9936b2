"""In-process interleaved A/B of GEMM tile variants on the step's real shapes (bf16). Run on the GPU box."""
import ctypes as C, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.manual_seed(0)
def bench(name, lay, M, N, K, variants, out_f32=1, split=0, rounds=7):
    T = torch.bfloat16
    K0, K = K, (K + 63) // 64 * 64          # the engine rounds K up over zero-padded buffers
    if lay == L.GEMM_NT: A = torch.randn(M, K, device='cuda').to(T); B = torch.randn(N, K, device='cuda').to(T); A[:, K0:] = 0; B[:, K0:] = 0; lda, ldb = K, K
    elif lay == L.GEMM_TN:
        Np = (N + 63) // 64 * 64            # the engine's buffers: leading dimension padded to 64 elements
        A = torch.randn(K, M, device='cuda').to(T); B = torch.randn(K, Np, device='cuda').to(T); A[K0:] = 0; B[K0:] = 0; lda, ldb = M, Np
    else:
        Np = (N + 63) // 64 * 64
        A = torch.randn(M, K, device='cuda').to(T); B = torch.randn(K, Np, device='cuda').to(T); A[:, K0:] = 0; B[K0:] = 0; lda, ldb = K, Np
    Cc = torch.zeros(M, N, device='cuda', dtype=torch.float32 if out_f32 else T)
    res = {v: [] for v in variants}
    ref = None
    for v in variants:                      # correctness of every variant against the first one
        Cc.zero_()
        a = L.GemmArgs(L.BF16, lay, A.data_ptr(), lda, B.data_ptr(), ldb, Cc.data_ptr(), N, M, N, K, 0, 0, None, 0, 0, 0, 0, out_f32, 0, 1.0, None, 1, v, split)
        L.check(lib.vmmt_gemm(C.byref(a), None), "g"); torch.cuda.synchronize()
        if ref is None: ref = Cc.float().clone()
        else:
            err = ((Cc.float() - ref).norm() / ref.norm()).item()
            if err > 2e-3: print("   !! variant %d differs from %d: rel %.3e" % (v, variants[0], err))
    for r in range(rounds):
        for v in variants:
            a = L.GemmArgs(L.BF16, lay, A.data_ptr(), lda, B.data_ptr(), ldb, Cc.data_ptr(), N, M, N, K, 0, 0, None, 0, 0, 0, 0, out_f32, 0, 1.0, None, 1, v, split)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
            s.record()
            for _ in range(5): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 5 * 1e3)
    fl = 2.0 * M * N * K
    print("%-28s" % name, "  ".join("%d: %7.1f us (%5.0f TF)" % (v, sorted(t)[len(t)//2], fl / sorted(t)[len(t)//2] / 1e6) for v, t in res.items()))
if __name__ == "__main__":
    V = [64, 128]
    bench("q fc1 NT 256x256x512", L.GEMM_NT, 256, 256, 512, V, out_f32=0)
    bench("q fc2 NT 256x256x256", L.GEMM_NT, 256, 256, 256, V, out_f32=1)
    bench("h1v NT 256x2048x256", L.GEMM_NT, 256, 2048, 256, V, out_f32=0)
    bench("mu_v NT 256x2048x2048", L.GEMM_NT, 256, 2048, 2048, V, out_f32=1)
    bench("dh1v NN 256x2048x2048", L.GEMM_NN, 256, 2048, 2048, V, out_f32=0)
    bench("dzt NN 256x256x2048", L.GEMM_NN, 256, 256, 2048, V, out_f32=1)
    bench("emb NN 5120x512x2048", L.GEMM_NN, 5120, 512, 2048, V, out_f32=1)
    bench("dW2 TN 2048x2048x256", L.GEMM_TN, 2048, 2048, 256, V, out_f32=1)
    bench("dWq TN 256x512x256", L.GEMM_TN, 256, 512, 256, V, out_f32=1)
    bench("Q NT 5120x512x512", L.GEMM_NT, 5120, 512, 512, V, out_f32=0)
    bench("AH NT 5120x512x1024", L.GEMM_NT, 5120, 512, 1024, V, out_f32=0)
