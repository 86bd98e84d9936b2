"""interleaved in-process A/B of scheduling options (step time, median over rounds)"""
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
configs = {
    "default-stream": dict(split_optim=True, bg=0, hi=False, side=True),
    "hi-priority main": dict(split_optim=True, bg=0, hi=True, side=True),
    "no aux stream": dict(split_optim=True, bg=0, hi=False, side=True, aux=False),
}
res = {k: [] for k in configs}
for r in range(5):
    for k, c in configs.items():
        eng.split_optim, eng.bg_adam_blocks, eng.use_side_stream = c["split_optim"], c["bg"], c["side"]
        eng.use_aux_stream = c.get("aux", True)
        torch.cuda.synchronize()
        ctx = torch.cuda.stream(eng.compute_stream) if c["hi"] else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            for i in range(4): step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20): step(i)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    v = sorted(v)
    print("%-26s median %.3f ms  min %.3f  max %.3f" % (k, v[len(v) // 2], v[0], v[-1]))
