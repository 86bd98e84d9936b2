"""Does the leading dimension of the generator's operands matter?  [V][512] bf16 rows are 1 KiB apart: every row segment of a K slab
has the same address bits 7..9.  Compare ld = 512 with padded leading dimensions (same data).  GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.manual_seed(0)
M, V, H = 5120, 30000, 512
T = torch.bfloat16
bias = torch.randn(V, device='cuda') * 0.1
y = torch.randint(4, V, (M,), device='cuda')
npart = lib.vmmt_gen_npart(V)
pm = torch.zeros(npart * M, device='cuda'); ps = torch.zeros_like(pm); pi = torch.zeros(npart * M, device='cuda', dtype=torch.int32)
tl = torch.zeros(M, device='cuda'); lse = torch.zeros(M, device='cuda'); nll = torch.zeros(M, device='cuda'); st = torch.zeros(8, device='cuda')
GT = torch.zeros(V, M, device='cuda', dtype=T)
W0 = (torch.randn(V, H, device='cuda') * 0.05).to(T); O0 = torch.randn(M, H, device='cuda').to(T)
fl = 2.0 * M * V * H
for variant in (3, 5):
    lib.vmmt_gen_set_variant(variant)
    for ld in (512, 520, 544, 576, 640):
        Wb = torch.zeros(V + 128, ld, device='cuda', dtype=T); Wb[:V, :H] = W0
        Ob = torch.zeros(M + 256, ld, device='cuda', dtype=T); Ob[:M, :H] = O0
        def fwd(): L.check(lib.vmmt_gen_loss_fwd(L.BF16, Wb.data_ptr(), ld, bias.data_ptr(), Ob.data_ptr(), ld, y.data_ptr(), M, V, H, 1, pm.data_ptr(), ps.data_ptr(), pi.data_ptr(), tl.data_ptr(), lse.data_ptr(), nll.data_ptr(), st.data_ptr(), None), "f")
        def bwd(): L.check(lib.vmmt_gen_loss_bwd(L.BF16, Wb.data_ptr(), ld, bias.data_ptr(), Ob.data_ptr(), ld, y.data_ptr(), M, V, H, 1, lse.data_ptr(), 1.0 / 256, GT.data_ptr(), M, None), "b")
        out = []
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            ts = []
            for r in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(); fn(); s.record()
                for _ in range(5): fn()
                e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / 5 * 1e3)
            t = sorted(ts)[2]
            out.append("%s %6.1f us (%4.0f TF)" % (name, t, fl / t / 1e6))
        print("variant %d ld %d: %s" % (variant, ld, "  ".join(out)), "lse sum %.4f" % lse.sum().item())
