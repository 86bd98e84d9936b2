"""host time per plan-entry kind for one training step (wraps the plan's callables)"""
import sys, time, torch, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench
cond = len(sys.argv) > 1 and sys.argv[1] == "cond"
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5, conditional=cond)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
eng.set_image_table(torch.rand(29000, d.img))
bs = bench.make_batches(d, 256, 20, 21, 29000, 4, "cuda", 1)
tl = torch.full((256,), 21, dtype=torch.int64, device="cuda")
def step(i):
    src, sl, tgt, idx = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True, tgt_len=tl if cond else None)
    eng.loss_backward(ws, normalization=256, batch_global=256)
    eng.optim_step()
    return ws
for i in range(5): ws = step(i)
torch.cuda.synchronize()
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(plan):
    for j, (fn, args, name, keep, sid) in enumerate(plan):
        if fn is None: continue
        def mk(fn, name):
            def w(*a):
                t0 = time.perf_counter(); r = fn(*a); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
            return w
        plan[j] = (mk(fn, name), args, name, keep, sid)
for pl in (ws.plan_fwd_train, ws.plan_loss_train, ws.plan_bwd): wrap(pl)
N = 10
t0 = time.perf_counter()
for i in range(N): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue %.3f ms/step" % ((t1 - t0) / N * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:12]:
    print("%-28s %8.1f us/step  (%d calls/step, %.1f us each)" % (k, v / N * 1e6, cnt[k] // N, v / cnt[k] * 1e6))
