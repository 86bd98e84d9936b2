"""the generator's weight-gradient product alone at the bench shape: dWg[V x H] = P[M x V]^T O'_slice[M x H] with the weighted column
sums (bias gradient) from the same pass -- vmmt_gemm TN, M_out = V = 30000, N = 512, K = 5120 tokens, one B operand per vocabulary slice
(b_batch_rows = 5120), against torch.matmul on the same operands.  python tools/dwg_one.py [nocs] [h1024]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
T = torch.bfloat16
g = torch.Generator().manual_seed(0)
H, K = (1024, 16384) if "h1024" in sys.argv else (512, 5120)
V = 50000 if "h1024" in sys.argv else 30000
ns, vps, mpad = C.c_int(), C.c_int(), C.c_int64()
L.check(lib.vmmt_gen_fused_geometry(K, V, H, C.byref(ns), C.byref(vps), C.byref(mpad)), "geometry")
ns, r = ns.value, vps.value
lda = (V + 31) // 32 * 32
P = torch.rand(K, lda, generator=g).to(T).cuda()
Os = (torch.randn(ns, K, H, generator=g) * 0.5).to(T).cuda()
w = torch.randn(ns, mpad.value, generator=g).cuda()
dW = torch.zeros(V, H, device="cuda"); db = torch.zeros(V, device="cuda")
a = L.GemmArgs()
a.dtype, a.layout, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = L.BF16, L.GEMM_TN, P.data_ptr(), lda, Os.data_ptr(), H, dW.data_ptr(), H, V, H, K
a.out_f32, a.alpha, a.split_k, a.b_batch_rows, a.b_batch_stride = 1, 1.0, 1, r, K * H
if "nocs" not in sys.argv:
    a.colsum_w, a.colsum_w_stride, a.colsum_out = w.data_ptr(), mpad.value, db.data_ptr()
    assert lib.vmmt_gemm_colsum_applies(C.byref(a)) == 1
f = lambda: L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
for _ in range(3): f()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
print("dWg %d x %d x %d, %d slices of %d rows%s: %.1f us = %.0f TFLOP/s" % (V, H, K, ns, r, "" if "nocs" in sys.argv else " + weighted column sums", us, 2.0 * V * H * K / us / 1e6))
if "check" in sys.argv:
    db.zero_(); f(); torch.cuda.synchronize()
    want = torch.cat([(P[:, i * r:min(V, (i + 1) * r)].float() * w[i, :K, None]).sum(0) for i in range(ns)])
    print("column sums: max rel err %.2e" % ((db - want).abs().max() / want.abs().max()).item())
Pt = P[:, :V].t()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for _ in range(3): torch.matmul(Pt[:r], Os[0])
t0.record()
for _ in range(20):
    for i in range(ns): torch.matmul(Pt[i * r:min(V, (i + 1) * r)], Os[i])
t1.record(); torch.cuda.synchronize()
print("torch.matmul, one call per slice: %.1f us" % (t0.elapsed_time(t1) / 20 * 1e3))
