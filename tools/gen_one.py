"""A few launches of the generator kernels (bf16, BASELINE config 2 shape) for counter collection.  GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.manual_seed(0)
M, V, H = 5120, 30000, 512
T = torch.bfloat16
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W = (torch.randn(V, H, device='cuda') * 0.05).to(T); O = torch.randn(M, H, device='cuda').to(T)
bias = torch.randn(V, device='cuda') * 0.1
y = torch.randint(4, V, (M,), device='cuda')
npart = lib.vmmt_gen_npart(V)
pm = torch.zeros(npart * M, device='cuda'); ps = torch.zeros_like(pm); pi = torch.zeros(npart * M, device='cuda', dtype=torch.int32)
tl = torch.zeros(M, device='cuda'); lse = torch.zeros(M, device='cuda'); nll = torch.zeros(M, device='cuda'); st = torch.zeros(8, device='cuda')
GT = torch.zeros(V, M, device='cuda', dtype=T)
lib.vmmt_gen_set_variant(variant)
for _ in range(6):
    L.check(lib.vmmt_gen_loss_fwd(L.BF16, W.data_ptr(), H, bias.data_ptr(), O.data_ptr(), H, y.data_ptr(), M, V, H, 1, pm.data_ptr(), ps.data_ptr(), pi.data_ptr(), tl.data_ptr(), lse.data_ptr(), nll.data_ptr(), st.data_ptr(), None), "f")
    L.check(lib.vmmt_gen_loss_bwd(L.BF16, W.data_ptr(), H, bias.data_ptr(), O.data_ptr(), H, y.data_ptr(), M, V, H, 1, lse.data_ptr(), 1.0 / 256, GT.data_ptr(), M, None), "b")
torch.cuda.synchronize()
