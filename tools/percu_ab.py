"""A/B of the one-workgroup-per-CU side-stream GEMMs (separate engines, interleaved rounds). GPU box only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5)
bs = bench.make_batches(d, 256, 20, 21, 29000, 4, "cuda", 1)
tab = torch.rand(29000, d.img)
engs = {}
for name, flag in (("two-per-CU", False), ("one-per-CU", True)):
    e = Engine(d, dtype="bf16", device="cuda", seed=0)
    e.side_one_per_cu = flag
    e.set_image_table(tab)
    engs[name] = e
def step(eng, i):
    src, sl, tgt, idx = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True)
    eng.loss_backward(ws, normalization=256, batch_global=256)
    eng.optim_step()
res = {k: [] for k in engs}
for r in range(5):
    for k, e in engs.items():
        for i in range(4): step(e, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20): step(e, i)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    v = sorted(v)
    print("%-12s median %.3f ms  min %.3f  max %.3f" % (k, v[len(v) // 2], v[0], v[-1]))
