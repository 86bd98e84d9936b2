"""host enqueue time of the data-parallel step through RCCL with a world of ONE rank (VMMT_DP_FORCE=1): which host calls of the
sharded optimiser path are expensive?   python tools/dp_host_time.py [sharded=1|0]"""
import os
import sys
import time

import torch

os.environ.setdefault("VMMT_DP_FORCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from variational_mmt_amd.dp import GradSync
from variational_mmt_amd.engine import Dims, Engine
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B = 256
cf = bench.CONFIGS["2"]
d = Dims(vs=cf["vs"], vt=cf["vt"], emb=cf["emb"], hid=cf["hid"], z=cf["z"], img=cf["img"], layers=cf["layers"], brnn=cf["brnn"], dropout=0.5)
eng = Engine(d, dtype="bf16", device=dev, seed=0)
eng.set_image_table(torch.rand(cf["n_img"], d.img))
bs = bench.make_batches(d, B, cf["S"], cf["T"], cf["n_img"], 4, dev, 1)
sync = GradSync(eng, sharded=(sys.argv[1] if len(sys.argv) > 1 else "1") == "1")
T = {"fwd": 0.0, "bwd": 0.0, "wait": 0.0, "opt": 0.0}


def step(i, acc=False):
    src, sl, tgt, idx, _tl, _ntok = bs[i % 4]
    t0 = time.perf_counter()
    ws = eng.forward(src, sl, tgt, idx, training=True)
    t1 = time.perf_counter()
    eng.loss_backward(ws, normalization=B, batch_global=B)
    t2 = time.perf_counter()
    sync.all_reduce()
    t3 = time.perf_counter()
    eng.optim_step()
    t4 = time.perf_counter()
    if acc:
        T["fwd"] += t1 - t0; T["bwd"] += t2 - t1; T["wait"] += t3 - t2; T["opt"] += t4 - t3


for i in range(6):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    step(i, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("sharded=%s: host enqueue %.3f ms/step, wall clock %.3f ms/step; host per phase (ms): %s" %
      (sync.sharded, (t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3, {k: round(v / 40 * 1e3, 3) for k, v in T.items()}))
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
dist.destroy_process_group()
