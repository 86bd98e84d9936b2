"""prints every event record / wait of the config-2 step's launch plans with its stream (0 main, 1 side, 2 aux, 3 tgt) and, between them, how
many kernels each stream got: which waits the main stream really needs.  python tools/dump_sync.py"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from variational_mmt_amd.engine import Dims, Engine

cf = bench.CONFIGS["2"]
d = Dims(vs=cf["vs"], vt=cf["vt"], emb=cf["emb"], hid=cf["hid"], z=cf["z"], img=cf["img"], layers=cf["layers"], brnn=cf["brnn"], dropout=0.5)
dev = torch.device("cuda:0")
eng = Engine(d, dtype="bf16", device=dev, seed=0)
eng.set_image_table(torch.rand(1000, d.img))
b = bench.make_batches(d, 256, cf["S"], cf["T"], 1000, 1, dev, 1, ragged=False)[0]
eng.hold_back = True
for _ in range(2):
    ws = eng.forward(b[0], b[1], b[2], b[3], training=True, n_tgt_tokens=b[5])
    eng.loss_backward(ws, normalization=256, batch_global=256)
    eng.optim_step(lr=0.002, max_grad_norm=5.0)
torch.cuda.synchronize()
for pname in ("plan_fwd_train", "plan_loss_train", "plan_bwd"):
    print("==", pname)
    cnt = [0, 0, 0, 0]
    for fn, args, name, keep, sid in getattr(ws, pname):
        if fn is None:
            print("   kernels since last sync entry per stream:", cnt)
            cnt = [0, 0, 0, 0]
            print("%-10s stream %d  %s" % (name, sid, args if isinstance(args, str) else ""))
        else:
            cnt[sid] += 1
    print("   kernels since last sync entry per stream:", cnt)
