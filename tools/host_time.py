import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
eng.set_image_table(torch.rand(29000, d.img))
bs = bench.make_batches(d, 256, 20, 21, 29000, 4, "cuda", 1)
def step(i):
    src, sl, tgt, idx, _tl = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True)
    eng.loss_backward(ws, normalization=256, batch_global=256)
    eng.optim_step()
for i in range(5): step(i)
torch.cuda.synchronize()
for use_side in (True, False):
    eng.use_side_stream = use_side
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("side=%s host enqueue %.3f ms/step, total %.3f ms/step" % (use_side, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
