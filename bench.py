#!/usr/bin/env python
"""Benchmark of the hot path: one VI_Model1 training step (image-row gather -> forward -> ELBO -> backward ->
[gradient all-reduce] -> clip + Adam) on synthetic Multi30k-shaped batches, BASELINE.json config 2 per GPU:
batch 256, src/tgt length 20, 30k vocabularies, 1-layer biLSTM 512, z 256, emb 500 (reference default), 2048-d
image features, bf16 compute / fp32 accumulate, dropout 0.5.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (metric = triplets/sec, whole job).  `roofline` is for the dominant kernel (timed live
with events on the launch stream); `cpu_baseline` is the CPU oracle timed on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def flops_per_triplet(d, S, Tp):
    """algorithmic forward FLOPs per triplet; a training step = 3x."""
    H, E, Z, V, D, L, dirs = d.hid, d.emb, d.z, d.vt, d.img, d.layers, d.dirs
    enc = S * sum(8 * H * ((E if l == 0 else H) + H // dirs) for l in range(L))
    dec = Tp * sum(8 * H * ((E + Z if l == 0 else H) + H) for l in range(L))
    att = Tp * (2 * H * H + 4 * S * H + 4 * H * H)
    gen = Tp * 2 * H * V
    qn = 2 * (2 * H * Z + 2 * Z * Z)
    im = 2 * Z + 2 * Z * D + 2 * D * D
    return dict(enc=enc, dec=dec, att=att, gen=gen, qnet=qn, img=im, total=enc + dec + att + gen + qn + im)


def make_batches(d, B, S, T, n_img, n_batches, device, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        src = torch.randint(2, d.vs, (S, B), generator=g)
        tgt = torch.randint(4, d.vt, (T, B), generator=g)
        tgt[0] = 2
        tgt[T - 1] = 3
        sl = torch.full((B,), S, dtype=torch.int64)
        idx = torch.randint(0, n_img, (B,), generator=g)
        out.append(tuple(x.to(device) for x in (src, sl, tgt, idx)))
    return out


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (oracle/vi1_oracle.py, validated against the real reference) timed on this host: BASELINE.json
    config 1 shape (batch 40), full step = forward + loss + backward + clip + Adam, dropout 0.5 masks included."""
    from oracle import vi1_oracle as O
    c = O.Cfg(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5)
    # torch's intra-op pool does not scale to hundreds of threads on these small matrices (256 threads: 150 s/step on
    # the MI355X host); 32 threads is near the best this CPU path reaches -- the count actually used is what is reported
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    p = O.init_params(c, seed=0)
    B, S, T = 40, 20, 21
    bt = O.synth_batch(c, B, S, T, n_img=512, seed=7)
    img = bt["table"][bt["indices"]]
    g = torch.Generator().manual_seed(3)
    state, steps, t0 = {}, 0, None
    while True:
        masks = {"dec_out": (torch.rand(T - 1, B, c.hid, generator=g) >= 0.5).float() * 2.0}
        if steps == 1:
            t0 = time.perf_counter()          # first step = warm-up
        r, Lo, gr = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], masks=masks)
        p, _ = O.clip_and_adam(p, gr, state)
        steps += 1
        if t0 is not None and (time.perf_counter() - t0 > seconds_budget or steps >= 31):
            break
    dt = time.perf_counter() - t0
    n = steps - 1
    return dict(value=round(B * n / dt, 2), unit="triplets/sec", cores=cores, kind="port",
                sample="%d full training steps of the CPU oracle (torch CPU fp32, %d threads) at batch 40, src/tgt len 20, "
                       "V=30000, 1-layer biLSTM 512, z 256 (BASELINE config 1), %.1f s" % (n, cores, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-stream", action="store_true", help="profiling aid: issue the whole step on one stream")
    ap.add_argument("--conditional", action="store_true", help="the --conditional prior variant (SURVEY.md 8f-1) instead of the fixed prior")
    ap.add_argument("--gen-variant", type=int, default=-1, help="generator main-loop variant (experiments; -1 = library default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, "--gpus must equal WORLD_SIZE (launch N > 1 with torch.distributed.run)"
    # rehearsal knobs (never set by the driver): VMMT_BENCH_ONE_GPU=1 puts every rank on cuda:0 and VMMT_BENCH_BACKEND=gloo
    # replaces RCCL, so that the N > 1 code path can be run end to end on a one-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("VMMT_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VMMT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from variational_mmt_amd.engine import Dims, Engine
    from variational_mmt_amd import _lib as L
    d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=a.dropout, conditional=a.conditional)
    B, S, T = a.batch, 20, 21
    Tp = T - 1
    eng = Engine(d, dtype=a.dtype, device=dev, seed=0)
    if a.gen_variant >= 0:
        L.lib().vmmt_gen_set_variant(a.gen_variant)
    if a.no_side_stream:
        eng.use_side_stream = False
    n_img = 29000
    gt = torch.Generator().manual_seed(11)
    eng.set_image_table(torch.rand(n_img, d.img, generator=gt))
    batches = make_batches(d, B, S, T, n_img, 8, dev, 1234 + rank)
    Bg = B * world
    tlen = torch.full((B,), T, dtype=torch.int64, device=dev)
    from variational_mmt_amd.dp import GradSync
    sync = GradSync(eng)          # attaches itself to the engine when torch.distributed runs with > 1 rank

    ev = {}

    def step(i, timed=False):
        src, sl, tgt, idx = batches[i % len(batches)]
        ws = eng.forward(src, sl, tgt, idx, training=True, tgt_len=tlen if a.conditional else None)
        eng.loss_backward(ws, normalization=Bg, batch_global=Bg)
        sync.all_reduce()          # waits for the segment all-reduces the backward plan issued behind each segment
        eng.optim_step(lr=0.002, max_grad_norm=5.0)
        return ws

    # ---- dominant-kernel timing hooks: events on the launch stream around the generator kernels -----------------
    ws0 = eng.workspace(B, S, Tp)
    dom = {"gen_fwd": [], "gen_bwd": []}

    def wrap(plan, index, key):
        fn, args, name, keep, sid = plan[index]

        def timed_fn(*x):
            if not dom.get("on"):
                return fn(*x)
            s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*x)
            e_.record()
            dom[key].append((s, e_))
            return rc
        timed_fn.__name__ = name
        plan[index] = (timed_fn, args, name, keep, sid)

    wrap(ws0.plan_loss_train, 0, "gen_fwd")

    for i in range(a.warmup):
        step(i)
    # backward plan exists now: hook the gen_loss_bwd entry
    for j, (fn, args, name, keep, sid) in enumerate(ws0.plan_bwd):
        if name in ("vmmt_gen_loss_bwd", "vmmt_gen_loss_bwd_db"):
            wrap(ws0.plan_bwd, j, "gen_bwd")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dom["on"] = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ws = step(a.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom["on"] = False
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = eng.read_stats(ws, batch_global=B)

    if rank == 0:
        fl = flops_per_triplet(d, S, Tp)
        ms = dt / a.steps * 1e3
        value = Bg * a.steps / dt
        # dominant kernel = the fused vocabulary projection (+log-softmax+NLL) GEMM passes: 2*M*V*H FLOP per launch
        M = Tp * B
        gen_flop = 2.0 * M * d.vt * d.hid
        t_f = sum(s.elapsed_time(e_) for s, e_ in dom["gen_fwd"]) / max(1, len(dom["gen_fwd"]))      # ms (fwd incl. combine)
        t_b = sum(s.elapsed_time(e_) for s, e_ in dom["gen_bwd"]) / max(1, len(dom["gen_bwd"]))
        t_dom = max(t_f, t_b)
        ach = gen_flop / (t_dom * 1e-3) / 1e12 if t_dom > 0 else 0.0
        traffic = None      # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            if B == 256 and a.dtype == "bf16":
                traffic = tj["read_bytes"] + tj["write_bytes"]
        except Exception:
            traffic = None
        out = {
            "metric": "triplets/sec", "value": round(value, 1), "unit": "triplets/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 2: VI_Model1 training step, batch %d/GPU, src/tgt len 20, V=30000, "
                                   "1-layer biLSTM 512, z 256, emb 500, 2048-d image feats, dropout %.1f, Adam%s" % (B, a.dropout, ", --conditional prior" if a.conditional else ""),
                       "global_batch": Bg, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "kernel": "gen_kernel (vocab projection + log-softmax/NLL pass, slower of fwd/bwd)",
                         "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "ms_fwd": round(t_f, 4), "ms_bwd": round(t_b, 4),
                         "step_tflops": round(3 * fl["total"] * Bg / (dt / a.steps) / 1e12, 2)},
            "elbo_per_sentence": round(st["elbo"] / B, 4),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
