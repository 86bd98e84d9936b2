import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from variational_mmt_amd.engine import Dims, Engine
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5, conditional=True)
e = Engine(d, dtype="bf16", device="cuda", seed=0)
e.set_image_table(torch.rand(29000, d.img))
bs = bench.make_batches(d, 256, 20, 21, 29000, 2, "cuda", 1)
tlen = torch.full((256,), 21, dtype=torch.int64, device="cuda")
for i in range(3):
    src, sl, tgt, idx = bs[i % 2]
    ws = e.forward(src, sl, tgt, idx, training=True, tgt_len=tlen)
    e.loss_backward(ws, normalization=256, batch_global=256)
    e.optim_step()
torch.cuda.synchronize()
for k, s in enumerate(e.seq_syncs):
    w = s.cpu().tolist()
    masks = []
    print("launch %d: epoch %d err %d cross-xcd members (cumulative) %d" % (k, w[0], w[2], w[3]))
