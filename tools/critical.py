"""Which launches does the step's length depend on?  For every kernel entry of the three launch plans (forward, loss, backward): the step timed with
that ONE entry left out (its results are then stale or garbage -- the timing is not), against the step as it is.  A launch whose removal
shortens the step by its own duration is on the critical path; one whose removal changes nothing is hidden behind something else.
    python tools/critical.py [config] [batch]          (one GPU box; ~1 minute)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cf = bench.CONFIGS[cfg]
d = Dims(vs=cf["vs"], vt=cf["vt"], emb=cf["emb"], hid=cf["hid"], z=cf["z"], img=cf["img"], layers=cf["layers"], brnn=cf["brnn"], dropout=0.5)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
eng.set_image_table(torch.rand(29000, d.img))
bs = bench.make_batches(d, BATCH, cf["S"], cf["T"], 29000, 4, "cuda", 1)
eng.hold_back = True


def step(i):
    src, sl, tgt, idx, _tl, ntok = bs[i % 4]
    ws = eng.forward(src, sl, tgt, idx, training=True, n_tgt_tokens=ntok)
    eng.loss_backward(ws, normalization=BATCH, batch_global=BATCH)
    eng.optim_step()
    return ws


def timed(n=24):
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        step(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for i in range(8):
    ws = step(i)
base = min(timed(), timed())
print("step as it is: %.1f us   (config %s, batch %d)" % (base, cfg, BATCH))
rows = []
# GROUP="bwd:15-27": these entries of one plan left out TOGETHER (what a fused kernel for that stretch could save at most)
grp = os.environ.get("GROUP")
if grp:
    pn, rng = grp.split(":")
    lo, hi = (int(x) for x in rng.split("-"))
    plan = getattr(ws, "plan_" + pn if pn.startswith("bwd") else "plan_" + pn)
    saved = {}
    for j in range(lo, hi + 1):
        fn, args, name, keep, sid = plan[j]
        if fn is not None:
            saved[j] = plan[j]
            plan[j] = ((lambda *x: 0), args, name, keep, sid)
    t = min(timed(), timed())
    for j, en in saved.items():
        plan[j] = en
    print("without %s entries %d..%d (%s): %.1f us, saves %.1f us" % (pn, lo, hi, ", ".join(sorted(set(e[2] for e in saved.values()))), t, base - t))
    sys.exit(0)
for pname in ("plan_fwd_train", "plan_loss_train", "plan_bwd"):
    plan = getattr(ws, pname)
    for j, entry in enumerate(plan):
        fn, args, name, keep, sid = entry
        if fn is None:
            continue
        plan[j] = ((lambda *x: 0), args, name, keep, sid)
        t = timed(16)
        plan[j] = entry
        rows.append((base - t, pname[5:], j, sid, name))
# the optimiser step's launches are not plan entries: the k-th vmmt_adam_step / vmmt_pack_multi call of a step left out in turn
for fname in ("vmmt_adam_step", "vmmt_pack_multi", "vmmt_sumsq"):
    real = getattr(eng.lib, fname)
    calls = [0]

    def counting(*x, _real=real, _c=calls):
        _c[0] += 1
        return _real(*x)
    setattr(eng.lib, fname, counting)
    step(0)
    per_step = calls[0]
    for k in range(per_step):
        state = [0]

        def skipping(*x, _real=real, _s=state, _k=k, _n=per_step):
            i = _s[0] % _n
            _s[0] += 1
            return 0 if i == _k else _real(*x)
        setattr(eng.lib, fname, skipping)
        t = timed(16)
        rows.append((base - t, "optim", k, -1, "%s call %d of %d" % (fname, k, per_step)))
    setattr(eng.lib, fname, real)
base2 = timed()
print("step as it is, measured again at the end: %.1f us" % base2)
print("%8s  %-10s %4s %6s  %s" % ("saves us", "plan", "#", "stream", "entry"))
for dlt, pn, j, sid, name in sorted(rows, reverse=True):
    print("%8.1f  %-10s %4d %6d  %s" % (dlt, pn, j, sid, name))
