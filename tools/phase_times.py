"""Main-stream phase timing WITHOUT the profiler: timing events recorded at every change of kernel name on the main
stream of a steady-state step (host far ahead of the GPU). Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench
cf = bench.CONFIGS[os.environ.get("CONFIG", "2")]          # CONFIG=script BATCH=40: the run scripts' own shape and batch
BATCH = int(os.environ.get("BATCH", "256"))
d = Dims(vs=cf["vs"], vt=cf["vt"], emb=cf["emb"], hid=cf["hid"], z=cf["z"], img=cf["img"], layers=cf["layers"], brnn=cf["brnn"], dropout=0.5)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
eng.set_image_table(torch.rand(29000, d.img))
eng.use_side_stream = os.environ.get("SIDE", "1") == "1"
bs = bench.make_batches(d, BATCH, cf["S"], cf["T"], 29000, 4, "cuda", 1)
sync = None
if os.environ.get("VMMT_DP_FORCE") == "1":       # the data-parallel step through RCCL with a world of one rank (SHARDED=0: replicated update)
    import torch.distributed as dist
    from variational_mmt_amd.dp import GradSync
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sync = GradSync(eng, sharded=os.environ.get("SHARDED", "1") == "1")
if os.environ.get("COARSE") == "1":      # events only where a recurrence, the sweep or the update begins / ends: ~10 events (~0.05 ms) instead of ~40
    eng.trace_only = {"vmmt_lstm_seq_fwd", "vmmt_lstm_seq_bwd", "vmmt_qnet_fwd", "vmmt_gen_fwd_dO", "vmmt_gen_fwd_combine", "vmmt_act_bwd", "vmmt_attn_fwd",
                      "vmmt_scatter_add_rows", "SUMSQ"}
eng.hold_back = os.environ.get("HOLD", "0") == "1"       # HOLD=1: as inside a training loop (the side-stream half of an update may be held back)
def step(i):
    src, sl, tgt, idx, _tl, _ntok = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True)
    eng.loss_backward(ws, normalization=BATCH, batch_global=BATCH)
    if sync is not None:
        sync.all_reduce()
    eng.optim_step()
for i in range(8): step(i)
torch.cuda.synchronize()
acc = {}
N = 6
for r in range(N):
    for i in range(3): step(i)          # let the host run ahead
    eng.trace = []
    step(3)
    tr, eng.trace = eng.trace, None
    end = torch.cuda.Event(enable_timing=True); end.record()
    torch.cuda.synchronize()
    tr.append(("END", end))
    for k in range(len(tr) - 1):
        key = (k, tr[k][0])
        acc.setdefault(key, []).append(tr[k][1].elapsed_time(tr[k + 1][1]) * 1e3)
tot = 0.0
for (k, name), v in sorted(acc.items()):
    m = sorted(v)[len(v) // 2]
    tot += m
    print("%3d %-28s %8.1f us   (cum %8.1f)" % (k, name, m, tot))
