"""host enqueue time per training step against the step's wall clock (is the host or the GPU the bottleneck?):
python tools/host_time.py [config=2|script] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cf = bench.CONFIGS[cfg]
d = Dims(vs=cf["vs"], vt=cf["vt"], emb=cf["emb"], hid=cf["hid"], z=cf["z"], img=cf["img"], layers=cf["layers"], brnn=cf["brnn"], dropout=0.5)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
eng.set_image_table(torch.rand(cf["n_img"], d.img))
bs = bench.make_batches(d, B, cf["S"], cf["T"], cf["n_img"], 4, "cuda", 1)


def step(i):
    src, sl, tgt, idx, _tl, _ntok = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True)
    eng.loss_backward(ws, normalization=B, batch_global=B)
    eng.optim_step()


for i in range(6):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("config %s batch %d: host enqueue %.3f ms/step, wall clock %.3f ms/step" % (cfg, B, (t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3))
# the host alone: the same 40 steps with every launch replaced by a no-op would need the plans' Python loop only
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(6)
