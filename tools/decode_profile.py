"""kernel profile of step-wise decoding: rocprofv3 --kernel-trace --stats -- python3 tools/decode_profile.py <batch> <beam>  (beam 1 = arg-max decoding)"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from variational_mmt_amd.engine import Dims, Engine
from variational_mmt_amd.decode import beam_decode, greedy_decode
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.0)
e = Engine(d, dtype="bf16", device="cuda", seed=0)
g = torch.Generator().manual_seed(0)
S, L = 20, 24
B, K = int(sys.argv[1]), int(sys.argv[2])
src = torch.randint(2, d.vs, (S, B), generator=g)
sl = torch.full((B,), S, dtype=torch.int64)
for _ in range(6):
    if K == 1:
        greedy_decode(e, src, sl, max_len=L)
    else:
        beam_decode(e, src, sl, K, max_len=L)
torch.cuda.synchronize()
