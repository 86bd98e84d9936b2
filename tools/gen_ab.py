"""In-process interleaved A/B of the generator kernel's main-loop variants (bf16, BASELINE config 2 shape). GPU box only."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.manual_seed(0)
M, V, H = 5120, 30000, 512
T = torch.bfloat16
W = (torch.randn(V, H, device='cuda') * 0.05).to(T); O = torch.randn(M, H, device='cuda').to(T)
bias = torch.randn(V, device='cuda') * 0.1
y = torch.randint(4, V, (M,), device='cuda')
npart = lib.vmmt_gen_npart(V)
pm = torch.zeros(npart * M, device='cuda'); ps = torch.zeros_like(pm); pi = torch.zeros(npart * M, device='cuda', dtype=torch.int32)
tl = torch.zeros(M, device='cuda'); lse = torch.zeros(M, device='cuda'); nll = torch.zeros(M, device='cuda'); st = torch.zeros(8, device='cuda')
GT = torch.zeros(V, M, device='cuda', dtype=T)
def fwd(): L.check(lib.vmmt_gen_loss_fwd(L.BF16, W.data_ptr(), H, bias.data_ptr(), O.data_ptr(), H, y.data_ptr(), M, V, H, 1, pm.data_ptr(), ps.data_ptr(), pi.data_ptr(), tl.data_ptr(), lse.data_ptr(), nll.data_ptr(), st.data_ptr(), None), "f")
def bwd(): L.check(lib.vmmt_gen_loss_bwd(L.BF16, W.data_ptr(), H, bias.data_ptr(), O.data_ptr(), H, y.data_ptr(), M, V, H, 1, lse.data_ptr(), 1.0 / 256, GT.data_ptr(), M, None), "b")
ref = {}
for v in (0, 3, 7, 8):
    lib.vmmt_gen_set_variant(v); st.zero_(); fwd(); bwd(); torch.cuda.synchronize()
    cur = (lse.clone(), GT.float().clone())
    if v == 0: ref = cur
    else: print("variant %d: lse maxdiff %.3e, GT relL2 %.3e" % (v, (cur[0] - ref[0]).abs().max().item(), ((cur[1] - ref[1]).norm() / ref[1].norm()).item()))
res = {}
for r in range(7):
    for v in (0, 3, 7, 8):
        lib.vmmt_gen_set_variant(v)
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(); fn(); s.record()
            for _ in range(5): fn()
            e.record(); torch.cuda.synchronize()
            res.setdefault((v, name), []).append(s.elapsed_time(e) / 5 * 1e3)
fl = 2.0 * M * V * H
for k in sorted(res):
    t = sorted(res[k])[len(res[k]) // 2]
    print("variant %d %s: %7.1f us (%4.0f TF)" % (k[0], k[1], t, fl / t / 1e6))
