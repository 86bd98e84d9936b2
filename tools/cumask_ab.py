"""interleaved in-process A/B of CU masks for the side stream (step time, median over rounds)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
eng.set_image_table(torch.rand(29000, d.img))
bs = bench.make_batches(d, 256, 20, 21, 29000, 4, "cuda", 1)
def step(i):
    src, sl, tgt, idx = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True)
    eng.loss_backward(ws, normalization=256, batch_global=256)
    eng.optim_step()
W = 8
configs = {
    "none": None,
    "all-ones": [0xFFFFFFFF] * W,
    "7/8 (0x7f7f7f7f)": [0x7F7F7F7F] * W,
    "3/4 (0x77777777)": [0x77777777] * W,
    "3/4 low 192": [0xFFFFFFFF] * 6 + [0, 0],
    "1/2 (0x55555555)": [0x55555555] * W,
    "1/2 (0x0f0f0f0f)": [0x0F0F0F0F] * W,
    "1/2 low 128": [0xFFFFFFFF] * 4 + [0] * 4,
    "5/8 (0x1f1f1f1f)": [0x1F1F1F1F] * W,
}
streams = {}
for k, m in configs.items():
    eng.set_side_cu_mask(m)
    streams[k] = eng.side_stream
res = {k: [] for k in configs}
for r in range(5):
    for k in configs:
        eng.side_stream = streams[k]
        torch.cuda.synchronize()
        for i in range(4): step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20): step(i)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    v = sorted(v)
    print("%-22s median %.3f ms  min %.3f  max %.3f" % (k, v[len(v) // 2], v[0], v[-1]))
