"""prints every entry of the config-2 step's launch plans (name, stream: 0 main, 1 side, 2 aux, 3 tgt), in issue order.  VMMT_DP_FORCE=1: the
data-parallel plans (a one-rank RCCL world).  python tools/dump_plan.py"""
import os
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
sync = None
if os.environ.get("VMMT_DP_FORCE") == "1":
    import torch.distributed as dist
    from variational_mmt_amd.dp import GradSync
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sync = GradSync(eng, sharded=True)
for _ in range(2):
    ws = eng.forward(b[0], b[1], b[2], b[3], training=True, n_tgt_tokens=b[5])
    eng.loss_backward(ws, normalization=256, batch_global=256)
    if sync is not None:
        sync.all_reduce()
    eng.optim_step(lr=0.002, max_grad_norm=5.0)
torch.cuda.synchronize()
for pname in ("plan_fwd_train", "plan_loss_train", "plan_bwd"):
    print("==", pname)
    for k, (fn, args, name, keep, sid) in enumerate(getattr(ws, pname)):
        print("%3d  s%d  %-26s %s" % (k, sid, name, args if isinstance(args, (str, tuple)) and fn is None else ""))
