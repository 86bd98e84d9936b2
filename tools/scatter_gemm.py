"""The embedding-gradient GEMM of the encoder tail in isolation: dX = dgates [5120 x 2048] W_ih [2048 x 500], scattered by token id
into the [30000 x 500] f32 gradient (atomic epilogue) vs stored plainly.  Run on the GPU box."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.manual_seed(0)
M, N, K, V = 5120, 500, 2048, 30000
T = torch.bfloat16
A = torch.randn(M, K, device="cuda").to(T)
B = torch.zeros(K, 512, device="cuda", dtype=T); B[:, :N] = torch.randn(K, N, device="cuda").to(T)
# token ids with a Zipf-like skew (a few very frequent rows, like real text)
ids = (torch.rand(M, device="cuda") ** 6 * (V - 2)).long() + 2
G = torch.zeros(V, N, device="cuda")
Cd = torch.zeros(M, N, device="cuda")
def run(scatter, split=0, reps=20):
    a = L.GemmArgs(L.BF16, L.GEMM_NN, A.data_ptr(), K, B.data_ptr(), 512, (G if scatter else Cd).data_ptr(), N, M, N, K, 0, 0, None, 0, 0, 0,
                   L.ACT_NONE, 1, 0, 1.0, ids.data_ptr() if scatter else None, 1, 0, split)
    for _ in range(3): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): L.check(lib.vmmt_gemm(C.byref(a), None), "g")
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("plain store      %7.1f us" % run(False))
print("scatter (atomic) %7.1f us" % run(True))
print("unique rows: %d of %d" % (ids.unique().numel(), M))
