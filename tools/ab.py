"""In-process A/B of engine scheduling switches at the benchmark shape (same box, same process, alternating rounds).
usage (GPU box): python tools/ab.py name1:attr=val,attr=val name2:attr=val ...      e.g.  base: aux:use_aux_stream=1 step:persistent_lstm=0"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench

cfgs = []
for a in sys.argv[1:]:
    name, _, rest = a.partition(":")
    kv = {}
    for item in filter(None, rest.split(",")):
        k, _, v = item.partition("=")
        kv[k] = (v == "1") if v in ("0", "1") else v      # other values stay strings
    cfgs.append((name, kv))
COND = os.environ.get("AB_CONDITIONAL", "0") == "1"
SCRIPT = os.environ.get("AB_SCRIPT", "0") == "1"            # the run scripts' shape instead of BASELINE config 2
BATCH = int(os.environ.get("AB_BATCH", "256"))
if SCRIPT:
    d = Dims(vs=30000, vt=30000, emb=500, hid=500, z=500, img=2048, layers=2, brnn=False, dropout=0.5, conditional=COND)
else:
    d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5, conditional=COND)
table = torch.rand(29000, d.img)
bs = bench.make_batches(d, BATCH, 20, 21, 29000, 4, "cuda", 1)
engs = []
TLEN = torch.full((BATCH,), 21, dtype=torch.int64, device="cuda")
for name, kv in cfgs:
    e = Engine(d, dtype="bf16", device="cuda", seed=0)
    for k, v in kv.items():
        assert hasattr(e, k) or k.startswith('ab_'), k
        setattr(e, k, v)
    e.set_image_table(table)
    if engs:        # all arms on the SAME streams: a process has few hardware queues, and the streams of a second engine share them
        for attr in ("side_stream", "_side_stream_plain", "aux_stream", "tgt_stream", "compute_stream"):
            setattr(e, attr, getattr(engs[0], attr))
    engs.append(e)

def run(e, n):
    if getattr(e, "ab_hi_pri_main", False):
        with torch.cuda.stream(e.compute_stream):
            return run_(e, n)
    return run_(e, n)


def run_(e, n):
    for i in range(n):
        src, sl, tgt, idx, _tl, _ntok = bs[i % 4]
        ws = e.forward(src, sl, tgt, idx, training=True, tgt_len=TLEN if COND else None)
        e.loss_backward(ws, normalization=BATCH, batch_global=BATCH)
        e.optim_step()

for e in engs:
    run(e, 6)
torch.cuda.synchronize()
res = {n: [] for n, _ in cfgs}
for rnd in range(5):
    for (name, _), e in zip(cfgs, engs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(e, 20)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for name, _ in cfgs:
    v = sorted(res[name])
    print("%-24s median %.3f ms   min %.3f   max %.3f" % (name, v[len(v) // 2], v[0], v[-1]))
