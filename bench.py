#!/usr/bin/env python
"""Benchmark of the hot path: one VI_Model1 training step (image-row gather -> forward -> ELBO -> backward ->
[gradient all-reduce] -> clip + Adam) on synthetic batches.

    python bench.py --gpus N --steps K --warmup W [--config 2|5|script]

--config 2 (default; the configuration BASELINE.json's metric is quoted on): batch 256 per GPU, src/tgt length 20, 30k
vocabularies, 1-layer biLSTM 512, z 256, emb 500 (reference default), 2048-d image features, bf16 compute / fp32
accumulate, dropout 0.5.  --config 5 (roofline stress): src/tgt length 64, 50k vocabularies, 2-layer 1024, z 512.
--config script: the run scripts as written (run_translated_m30k_only.sh:46-57, opts.py defaults): 2-layer uni-directional LSTM
500, z 500, emb 500 -- sizes the engine computes padded to 512 (engine.Dims.hp).

N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks itself (`python -m torch.distributed.run ...
bench.py --gpus N ...`: RANK / WORLD_SIZE are in the environment) or a plain `python bench.py --gpus N` starts them: the
parent process then only spawns `torch.distributed.run` as a child BEFORE touching any GPU, relays rank 0's JSON line and
exits with the child's code.

Prints ONE JSON line on rank 0 (metric = triplets/sec, whole job).  `roofline` is for the dominant kernel (timed live
with events on the launch stream); `cpu_baseline` is the CPU oracle timed on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402   (importing torch does not initialise the GPU)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    "2": dict(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, S=20, T=21, n_img=29000,
            name="BASELINE config 2: VI_Model1 training step, batch %d/GPU, src/tgt len 20, V=30000, 1-layer biLSTM 512, z 256, "
                 "emb 500, 2048-d image feats"),
    "5": dict(vs=50000, vt=50000, emb=1024, hid=1024, z=512, img=2048, layers=2, brnn=True, S=64, T=65, n_img=1000000,
            name="BASELINE config 5 (roofline stress): VI_Model1 training step, batch %d/GPU, src/tgt len 64, V=50000, 2-layer "
                 "biLSTM 1024, z 512, emb 1024, 2048-d image feats"),
    "script": dict(vs=30000, vt=30000, emb=500, hid=500, z=500, img=2048, layers=2, brnn=False, S=20, T=21, n_img=29000,
                   name="run scripts as written (run_translated_m30k_only.sh): VI_Model1 training step, batch %d/GPU, src/tgt len 20, V=30000, "
                        "2-layer uni-directional LSTM 500, z 500, emb 500, 2048-d image feats"),
}


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


def make_batches(d, B, S, T, n_img, n_batches, device, seed, ragged=False):
    """seeded synthetic batches: src ids U[2, V), tgt ids U[4, V) with <s> first and </s> last, <blank> = 1
    beyond a sentence's length; lengths all S / T ("Multi30k-shaped, fixed"), or with `ragged` U[S/2, S] sorted by decreasing source
    length as the reference's iterator delivers them (tensor shapes = the longest sentence of the batch)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        if ragged:
            sl = torch.sort(torch.randint(S // 2, S + 1, (B,), generator=g), descending=True).values
            tl = torch.randint((T - 1) // 2, T, (B,), generator=g) + 1              # incl. <s> and </s>
            S_, T_ = int(sl.max()), int(tl.max())
        else:
            sl, tl, S_, T_ = torch.full((B,), S, dtype=torch.int64), torch.full((B,), T, dtype=torch.int64), S, T
        src = torch.randint(2, d.vs, (S_, B), generator=g)
        tgt = torch.randint(4, d.vt, (T_, B), generator=g)
        src[torch.arange(S_).unsqueeze(1) >= sl.unsqueeze(0)] = 1
        tgt[0] = 2
        tgt[tl - 1, torch.arange(B)] = 3
        tgt[torch.arange(T_).unsqueeze(1) >= tl.unsqueeze(0)] = 1
        idx = torch.randint(0, n_img, (B,), generator=g)
        # (the last item: decoder rows that carry a target -- what a loader knows on the host; Engine.forward(n_tgt_tokens=))
        out.append(tuple(x.to(device) for x in (src, sl, tgt, idx, tl)) + (int((tl - 1).sum()),))
    return out


def gpu_state():
    """what the box lets an ordinary user read about its GPUs without starting another program: per card, from sysfs, the active shader
    / memory clock (MHz), the socket power (W), the busy percentage.  A list with one entry per card that exposes a clock table."""
    import glob
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        d_ = os.path.dirname(f)
        ent = {"card": os.path.basename(os.path.dirname(d_))}

        def active(path):
            try:
                for ln in open(path):
                    if "*" in ln:
                        return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()) or 0)
            except (OSError, IndexError, ValueError):
                pass
            return None
        ent["sclk"], ent["mclk"] = active(f), active(os.path.join(d_, "pp_dpm_mclk"))
        for name, key, scale in (("power1_average", "watts", 1e-6), ("power1_input", "watts", 1e-6), ("temp1_input", "temp_c", 1e-3)):
            if key in ent:
                continue
            for h in glob.glob(os.path.join(d_, "hwmon", "hwmon*", name)):
                try:
                    ent[key] = round(int(open(h).read().strip()) * scale, 1)
                except (OSError, ValueError):
                    pass
        try:
            ent["busy"] = int(open(os.path.join(d_, "gpu_busy_percent")).read().strip())
        except (OSError, ValueError):
            pass
        out.append(ent)
    return out


def host_sched():
    """what the kernel says about this thread's and this container's share of the host: ms this thread spent runnable but waiting for a
    CPU (/proc/thread-self/schedstat), involuntary context switches, and the container's CPU-quota throttling (cgroup cpu.stat)"""
    out = {}
    try:
        on_cpu, waited, _ = open("/proc/thread-self/schedstat").read().split()[:3]
        out["on_cpu_ms"], out["runq_wait_ms"] = int(on_cpu) / 1e6, int(waited) / 1e6
    except (OSError, ValueError):
        pass
    try:
        for ln in open("/proc/thread-self/status"):
            if ln.startswith("nonvoluntary_ctxt_switches"):
                out["preempted"] = int(ln.split()[1])
    except (OSError, ValueError, IndexError):
        pass
    try:            # the whole process (HIP's helper threads signal the synchronize's return)
        tot = 0
        for t_ in os.listdir("/proc/self/task"):
            tot += int(open("/proc/self/task/%s/schedstat" % t_).read().split()[1])
        out["runq_wait_all_threads_ms"] = tot / 1e6
    except (OSError, ValueError, IndexError):
        pass
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            for ln in open(f):
                k, _, v = ln.partition(" ")
                if k in ("nr_throttled", "throttled_usec", "throttled_time"):
                    out["cgroup_" + ("throttled_ms" if k != "nr_throttled" else k)] = int(v) / (1.0 if k == "nr_throttled" else 1e3 if k == "throttled_usec" else 1e6)
            break
        except (OSError, ValueError):
            continue
    return out


def sched_delta(a_, b_):
    return {k: round(b_[k] - a_[k], 3) for k in b_ if k in a_}


def cpu_quota():
    """the container's CPU quota in CPUs (cgroup v2 cpu.max / v1 cfs_quota), None = unlimited or unreadable"""
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p_), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        return None if q <= 0 else round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
    except (OSError, ValueError):
        return None


def step_summary(dev_ms, host_ms):
    """{min, median, p90, max, list} of the per-step device times of a timed region + the steps that stand out"""
    import statistics
    srt = sorted(dev_ms)
    med = statistics.median(srt)
    p90 = srt[min(len(srt) - 1, int(round(0.9 * (len(srt) - 1))))]
    slow = [{"step": i, "ms": round(x, 3), "host_ms": round(host_ms[i], 3)} for i, x in enumerate(dev_ms) if x > 1.5 * med]
    return {"min": round(srt[0], 4), "median": round(med, 4), "p90": round(p90, 4), "max": round(srt[-1], 4),
            "list": [round(x, 3) for x in dev_ms]}, slow, med


def diagnose(ms, steps, step_med, slow_steps, enq_ms, sched, reps, overlap):
    """one sentence: is the whole-region ms/step what the box does step after step, and if not, what took the rest (tests/test_bench_diagnosis.py)"""
    out = "steady: whole-region ms/step within 5 %% of the median step (%.3f ms)" % step_med
    if ms > 1.05 * step_med:
        lost = ms * steps - step_med * steps
        ends = sched.get("region_ms", ms * steps) - sched.get("device_span_ms", step_med * steps)
        if sched.get("cgroup_nr_throttled", 0) > 0 or sched.get("runq_wait_ms", 0.0) > 0.5 * lost:
            out = ("HOST DESCHEDULED: while it enqueued the region this thread waited %.1f ms for a CPU (%d preemptions) and the container's CPU quota "
                   "throttled it (%.1f ms summed over its threads, %d periods); the region lost %.1f ms against %d median steps of %.3f ms" %
                   (sched.get("runq_wait_ms", 0.0), sched.get("preempted", 0), sched.get("cgroup_throttled_ms", 0.0), sched.get("cgroup_nr_throttled", 0),
                    lost, steps, step_med))
        elif slow_steps:
            out = ("STALL: %d of %d steps took > 1.5 x the median step (%.3f ms) and account for %.1f of the %.1f ms the region "
                   "lost against %d median steps; " % (len(slow_steps), steps, step_med, sum(x["ms"] - step_med for x in slow_steps), lost, steps)) + \
                ("the host was late there (enqueue took longer than the step)" if any(x["host_ms"] > x["ms"] * 0.8 for x in slow_steps)
                 else "the host was ahead: the device itself stalled")
        elif enq_ms > 0.9 * ms * steps:
            out = "HOST-BOUND: enqueueing the region took %.1f of its %.1f ms" % (enq_ms, ms * steps)
        else:
            out = ("region %.3f ms/step against a median step of %.3f ms with no single slow step: the device ran the %d steps in %.1f ms, the region's "
                   "wall clock has %.1f ms more at its ends (first launch, the synchronize's return)" % (ms, step_med, steps, sched.get("device_span_ms", 0.0), ends))
    if overlap is not None and overlap["ratio"] <= 1.12:
        out += "; " + overlap["verdict"]
    if len(reps) > 1 and ms > 1.10 * min(reps[1:]):
        out += "; the same region repeated ran at %s ms/step: the official (first) region was NOT typical for this box" % ", ".join("%.3f" % x for x in reps[1:])
    return out


def host_cpu():
    """model name, physical cores, logical CPUs of the host (from /proc/cpuinfo; no extra tools needed)"""
    model, phys, logical = "?", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif k == "processor":
                logical += 1
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return model, len(phys) or None, logical or (os.cpu_count() or 1)


def _cpu_run(threads, seconds_budget, max_steps):
    """full training steps (forward + loss + backward + clip + Adam, dropout masks included) of the CPU oracle with torch's
    fused LSTM (oracle/fast_cpu.py = the reference's own CPU kernel choice) at BASELINE config 1: batch 40.
    -> (triplets/s, timed steps, seconds, [fwd, loss+bwd, optim] seconds per step)"""
    from oracle import fast_cpu as F
    from oracle import vi1_oracle as O
    c = O.Cfg(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.5)
    torch.set_num_threads(threads)
    p = O.init_params(c, seed=0)
    B, S, T = 40, 20, 21
    bt = O.synth_batch(c, B, S, T, n_img=512, seed=7)
    img = bt["table"][bt["indices"]]
    g = torch.Generator().manual_seed(3)
    state, steps, t0 = {}, 0, None
    phase = [0.0, 0.0, 0.0]
    first = None
    while True:
        masks = {"dec_out": (torch.rand(T - 1, B, c.hid, generator=g) >= 0.5).float() * 2.0}
        if steps == 1:
            t0 = time.perf_counter()          # first step = warm-up
            phase = [0.0, 0.0, 0.0]
        ta = time.perf_counter()
        pp = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
        r = O.forward(pp, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], True, masks, False, lstm_layer=F.lstm_layer)
        tb = time.perf_counter()
        Lo = O.loss(pp, c, r, bt["tgt"], img)
        (Lo["loss"] / float(B)).backward()
        gr = {k: v.grad.detach() for k, v in pp.items() if v.grad is not None}
        tc = time.perf_counter()
        p, _ = F.clip_and_adam(p, gr, state)          # in place, torch.optim.Adam's operation order (the reference's optimiser)
        td = time.perf_counter()
        phase[0] += tb - ta
        phase[1] += tc - tb
        phase[2] += td - tc
        steps += 1
        if steps == 1:
            first = td - ta
            if first > seconds_budget:        # a thread count at which ONE step already exceeds the budget: report that step
                return B / first, 1, first, [tb - ta, tc - tb, td - tc]
        if t0 is not None and (time.perf_counter() - t0 > seconds_budget or steps > max_steps):
            break
    dt = time.perf_counter() - t0
    n = steps - 1
    return B * n / dt, n, dt, [x / n for x in phase]


def cpu_baseline(seconds_budget=10.0):
    """The CPU oracle (oracle/vi1_oracle.py, validated against the real reference; LSTM through torch's fused CPU kernel like
    the reference) timed on this host at 8 threads (the core count the reference itself was timed on at survey time),
    32 threads, and ALL physical cores (torch's intra-op pool does not scale on these small matrices: the run
    is bounded to one step when a step takes longer than the budget).  `value` = the fastest run, `cores` = the threads it used;
    every run with its forward / loss+backward / optimiser split is listed in `runs`."""
    model, phys, logical = host_cpu()
    runs = {}
    counts = sorted({min(8, logical), min(32, logical), min(phys or logical, logical)})
    for th in counts:
        v, n, dt, ph = _cpu_run(th, seconds_budget if th <= 32 else 6.0, 40)
        runs[th] = dict(triplets_per_sec=round(v, 2), steps=n, seconds=round(dt, 2), fwd_s=round(ph[0], 4), loss_bwd_s=round(ph[1], 4),
                        optim_s=round(ph[2], 4))
    best = max(runs, key=lambda k: runs[k]["triplets_per_sec"])
    b = runs[best]
    return dict(value=b["triplets_per_sec"], unit="triplets/sec", cores=best, kind="port",
                host_cpu=model, host_physical_cores=phys, host_logical_cpus=logical,
                by_threads={str(k): runs[k]["triplets_per_sec"] for k in sorted(runs)}, runs={str(k): runs[k] for k in sorted(runs)},
                sample="%d full training steps of the CPU oracle (torch CPU fp32, fused ATen LSTM, %d threads) at batch 40, src/tgt "
                       "len 20, V=30000, 1-layer biLSTM 512, z 256 (BASELINE config 1), %.1f s; per step: forward %.3f s, loss + backward "
                       "%.3f s, clip + Adam %.3f s" % (b["steps"], best, b["seconds"], b["fwd_s"], b["loss_bwd_s"], b["optim_s"]))


def parity_gate(dev, dtype, dropout):
    """the parity figures that go with every number.  ONE training step at BASELINE config 1's shape (batch 40,
    src/tgt length 20, V = 30 000, 1-layer biLSTM 512, z 256 -- the cpu_baseline's own batch, `oracle.synth_batch(seed 7)`) on the HIP
    path in the benchmark's arithmetic (dtype, dropout), and the same step on the CPU oracle with the sample eps and the device's own
    dropout mask injected: |dELBO| / |ELBO|, |dKL| / |KL|, max |d per-token NLL| (reference statistics: onmt/VILoss.py:478-485).
    The oracle is the checker here, never the thing measured."""
    from oracle import vi1_oracle as O
    from variational_mmt_amd.engine import Dims, Engine
    c = O.Cfg(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=dropout)
    p = O.init_params(c, seed=0)
    B, S, T = 40, 20, 21
    bt = O.synth_batch(c, B, S, T, n_img=512, seed=7)
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, dropout), dtype=dtype, device=dev, seed=0)
    e.load_state_dict(p)
    e.set_image_table(bt["table"])
    ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
    e.loss_backward(ws, normalization=B)
    torch.cuda.synchronize()
    st = e.read_stats(ws)
    masks = {"dec_out": ws.out_mask.view().float().cpu().view(T - 1, B, c.hid)} if dropout > 0 else None
    tok = ws.tok_nll.float().cpu().view(T - 1, B)
    img = bt["table"][bt["indices"]]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        r = O.forward(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], True, masks, False)
        Lo = O.loss(p, c, r, bt["tgt"], img)
    elbo, kl = float(Lo["elbo"]), float(Lo["kl_before"])
    out = {"elbo_rel": abs(st["elbo"] - elbo) / abs(elbo), "kl_rel": abs(st["td_kl_before"] - kl) / abs(kl),
           "tok_nll_max_abs": float((tok - Lo["tok_nll"]).abs().max()), "nll_rel": abs(st["nmt"] - float(Lo["nll"])) / abs(float(Lo["nll"])),
           "elbo": round(st["elbo"], 4), "elbo_oracle": round(elbo, 4), "n_words": st["n_words"], "n_words_oracle": int(Lo["n_words"]),
           "vs": "oracle (CPU restatement pinned to the reference), cfg-1 shape, B 40, %s, dropout %.1f with the device's mask injected" % (dtype, dropout)}
    for k in ("elbo_rel", "kl_rel", "tok_nll_max_abs", "nll_rel"):
        out[k] = float("%.3g" % out[k])
    del e
    torch.cuda.empty_cache()
    return out


def through_trainer(a, dev, rank, world):
    """`--through-trainer`: the drop-in surface end to end.  A synthetic dataset of 29 000 triplets (Multi30k's size; source /
    target lengths U[10, 20], word ids Zipf-distributed over 30 k-word vocabularies) is walked by onmt.io.OrderedIterator (pools of 100
    batches sorted by length, shuffled batches: host numericalisation + H2D of the ids every step) and trained by
    onmt.TrainerMultimodal.train (ModelConstructor / VILoss / Optim mirrors) for two epochs; the SECOND epoch is timed (the first one
    builds the launch plans of every shape bucket).  Reports triplets/s over the epoch, the share of the wall clock the host needed to
    prepare and enqueue it (loop time minus the time the loader spent waiting for the device to catch up: < 1 means the GPU is the
    bottleneck), and the number of workspace shape buckets."""
    import random
    import types
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    from variational_mmt_amd.onmt.io import textdata as td
    cf = CONFIGS[a.config]
    V, N, B = cf["vs"], 29000, a.batch
    rng = random.Random(1234)
    itos_s = ["<unk>", "<blank>"] + ["s%d" % i for i in range(V - 2)]
    itos_t = ["<unk>", "<blank>", "<s>", "</s>"] + ["t%d" % i for i in range(V - 4)]
    fields = td.get_fields()
    fields["src"].vocab, fields["tgt"].vocab = td.Vocab(itos_s), td.Vocab(itos_t)
    # Zipf-ish ranks: id = floor(V ** u), u ~ U[0, 1): log-uniform over the vocabulary
    uniform_ids = os.environ.get("VMMT_TT_UNIFORM") == "1"          # diagnostic knobs: uniform word ids / all sentences full length
    fixed_len = os.environ.get("VMMT_TT_FIXED") == "1"

    def words(itos, lo, n):
        if uniform_ids:
            return tuple(itos[rng.randrange(lo, len(itos))] for _ in range(n))
        return tuple(itos[min(len(itos) - 1, lo + int((len(itos) - lo) ** rng.random()) - 1)] for _ in range(n))
    examples = []
    for i in range(N):
        ex = td.Example()
        # source 10..20 words; target 8..18 words + <s> + </s> = 10..20 positions: the lengths of `--lengths ragged` (make_batches)
        ls, lt = (20, 18) if fixed_len else (rng.randint(10, 20), rng.randint(8, 18))
        ex.src, ex.tgt, ex.indices = words(itos_s, 2, ls), words(itos_t, 4, lt), i
        examples.append(ex)
    ds = td.TextDataset(examples, fields)
    opt = types.SimpleNamespace(model_type="text", multimodal_model_type="vi-model1", path_to_train_img_feats="resnet50.hdf5",
                                src_word_vec_size=cf["emb"], tgt_word_vec_size=cf["emb"], rnn_size=cf["hid"], z_latent_dim=cf["z"],
                                enc_layers=cf["layers"], dec_layers=cf["layers"], encoder_type="brnn" if cf["brnn"] else "rnn",
                                brnn=cf["brnn"], dropout=a.dropout, param_init=0.1, gpuid=[dev.index], seed=1, compute_dtype=a.dtype,
                                conditional=a.conditional, rnn_type="LSTM", global_attention="general")
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, None)
    loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    gt = torch.Generator().manual_seed(11)
    feats = torch.rand(N, cf["img"], generator=gt).to(dev)
    trainer = onmt.TrainerMultimodal(model, loss, loss, optim, 0, 32, "text", "sents", 1, train_img_feats=feats, valid_img_feats=feats[:64],
                                     multimodal_model_type="vi-model1")
    def epoch(n):
        it = onmt.io.OrderedIterator(dataset=ds, batch_size=B, device=dev, sort=False, train=True, sort_within_batch=True, repeat=False,
                                     dp_rank=rank, dp_world=world) if world > 1 else \
            onmt.io.OrderedIterator(dataset=ds, batch_size=B, device=dev, sort=False, train=True, sort_within_batch=True, repeat=False)
        torch.cuda.synchronize()
        mark = {}
        real_check = eng.check_async_errors
        eng.check_async_errors = lambda: (mark.setdefault("t", time.perf_counter()), real_check())[1]    # (it synchronises: note when the loop got there)
        stg = td._staging.get(str(dev))
        w0 = stg.waited if stg is not None else 0.0
        t0 = time.perf_counter()
        st = trainer.train(it, n, None)
        eng.check_async_errors = real_check
        t_host = mark.get("t", time.perf_counter()) - t0          # every step enqueued ...
        stg = td._staging.get(str(dev))
        t_host -= (stg.waited - w0) if stg is not None else 0.0   # ... minus the time the loader waited for the device to catch up
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t_host, st
    eng = model.engine
    epoch(1)
    buckets = sum(1 for k in eng.ws if isinstance(k, tuple) and len(k) == 3)
    dt, t_host, st = epoch(2)
    steps = (N + B * world - 1) // (B * world)
    if rank == 0:
        out = {"metric": "triplets/sec", "value": round(N / dt, 1), "unit": "triplets/sec", "n_gpus": world, "steps": steps, "warmup": steps,
               "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype,
               "data": "synthetic",
               "config": {"workload": "through onmt.TrainerMultimodal.train + onmt.io.OrderedIterator: one epoch over 29000 synthetic triplets, "
                                      "lengths U[10,20], Zipf word ids, " + (cf["name"] % B) + ", dropout %.1f, Adam" % a.dropout,
                          "global_batch": B * world, "parallelism": "dp%d" % world},
               "host_busy_share": round(t_host / dt, 3), "shape_buckets": buckets, "workspace_evictions": eng.ws_evictions,
               "workspace_gb": round(eng.workspace_bytes() / 2 ** 30, 2),
               "train_ppl": round(st.ppl(), 2), "elbo_per_sentence": round(st.elbo_loss / N, 3)}
        emit(out)


_REAL_STDOUT = None


def quiet_stdout():
    """From here on everything that libraries print on stdout goes to stderr (RCCL prints a version banner on stdout when its first
    communicator comes up; MIOpen / the loader have their own lines): the run's stdout carries ONE line, the result (emit)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a):
    """parent of a plain `python bench.py --gpus N`: start the N ranks as a child `torch.distributed.run` (this process never
    touches a GPU and never re-execs), relay rank 0's JSON line, exit with the child's code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln or ln.startswith('{"selftest"'):
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if r.returncode != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank child failed (exit code %d)\n" % (a.gpus, r.returncode))
        sys.exit(r.returncode or 1)
    print(line, flush=True)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 50 x 1.8 ms: still a 0.1-second timed region; 20 steps read ~1 % slow
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS), help="BASELINE.json configuration (2 = headline, 5 = stress) or "
                    "'script' (the run scripts' own sizes: 2-layer uni-directional 500)")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions run back to back in the same process; the FIRST is the metric, the "
                    "others are reported beside it (`repeats_ms`) to show whether it was typical for the box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gate (one cfg-1-shaped step against the CPU oracle)")
    ap.add_argument("--no-side-stream", action="store_true", help="profiling aid: issue the whole step on one stream")
    ap.add_argument("--no-overlap-probe", action="store_true", help="profiling aid (tools/prof.sh, tools/pmc.sh): leave out the stream-overlap "
                    "self-check, whose 13 extra steps -- 6 of them on ONE stream -- would be averaged into a kernel profile of the run")
    ap.add_argument("--conditional", action="store_true", help="the --conditional prior variant (SURVEY.md 8f-1) instead of the fixed prior")
    ap.add_argument("--n-img", type=int, default=0, help="rows of the resident image-feature table (default: the configuration's; 290000 = "
                    "BASELINE config 4's 290 K-triplet set, 1000000 = config 5's synthetic 1 M triplets: 8.2 GB of HBM)")
    ap.add_argument("--lengths", default="fixed", choices=["fixed", "ragged"], help="sentence lengths: all 20 (the headline shape) or U[10, 20] "
                    "sorted by source length (BASELINE.md section 3)")
    ap.add_argument("--through-trainer", action="store_true", help="time onmt.TrainerMultimodal.train over an OrderedIterator on a synthetic "
                    "29 K-triplet dataset (the drop-in surface: host numericalisation, H2D of ids, ragged lengths, shape buckets) instead of "
                    "the engine-level step")
    ap.add_argument("--selftest-launch", action="store_true", help="CPU check of the N-rank launch path: gloo ranks, one all-reduce, no GPU")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(a)           # does not return

    quiet_stdout()
    # the host side of the product is ONE enqueueing thread; torch's CPU pool (one thread per core of the HOST, 128 here, spinning for
    # milliseconds after every parallel region) only serves the synthetic data and the parameter initialisation: inside a container with a
    # CPU quota that pool can spend the quota and get the enqueueing thread throttled with it (cpu_baseline sets its own thread counts)
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.selftest_launch:
        import torch.distributed as dist
        if os.environ.get("VMMT_SELFTEST_FAIL_RANK") == str(rank):
            sys.exit(3)               # tests: a failing rank must fail the whole invocation
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.destroy_process_group()
        if rank == 0:
            emit({"selftest": world})
        return
    # rehearsal knobs (never set by the driver): VMMT_BENCH_ONE_GPU=1 puts every rank on cuda:0 and VMMT_BENCH_BACKEND=gloo
    # replaces RCCL, so that the N > 1 code path can be run end to end on a one-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("VMMT_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world == 1 and os.environ.get("VMMT_DP_FORCE") == "1":
        # one-GPU rehearsal of the RCCL path: a process group of ONE rank on the `nccl` backend; dp.GradSync attaches to it and every
        # collective of the step runs through RCCL with itself as the only peer (the "dp" block of the line then shows real RCCL calls)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group(os.environ.get("VMMT_BENCH_BACKEND", "nccl"), rank=0, world_size=1, device_id=dev)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VMMT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if a.through_trainer:
        return through_trainer(a, dev, rank, world)
    from variational_mmt_amd.engine import Dims, Engine
    cf = CONFIGS[a.config]
    d = Dims(vs=cf["vs"], vt=cf["vt"], emb=cf["emb"], hid=cf["hid"], z=cf["z"], img=cf["img"], layers=cf["layers"], brnn=cf["brnn"],
             dropout=a.dropout, conditional=a.conditional)
    B, S, T = a.batch, cf["S"], cf["T"]
    Tp = T - 1
    eng = Engine(d, dtype=a.dtype, device=dev, seed=0)
    if a.no_side_stream:
        eng.use_side_stream = False
    if os.environ.get("VMMT_BENCH_ONE_GPU") == "1" and world > 1 and os.environ.get("VMMT_PERSISTENT_LSTM") is None:
        # rehearsal only: the persistent recurrence kernels need every workgroup of a launch resident at once, which two processes
        # sharing one GPU cannot both have (the launches would run into their 2-second hand-off bound, and the engine would fall
        # back to the per-step kernels by itself after two slow steps: Engine._seq_timeout_fallback)
        eng.persistent_lstm = False
    n_img = a.n_img or cf["n_img"]
    if n_img <= 100000:
        gt = torch.Generator().manual_seed(11)
        eng.set_image_table(torch.rand(n_img, d.img, generator=gt))
    else:           # a table of GBs is drawn on the device (no host copy of it): the reference's array is the HDF5 file's, read once
        gd = torch.Generator(device=dev).manual_seed(11)
        tab = torch.empty(n_img, d.img, dtype=torch.float32, device=dev)
        for lo in range(0, n_img, 65536):
            tab[lo:lo + 65536].uniform_(0.0, 1.0, generator=gd)
        eng.set_image_table(tab)
        del tab
    batches = make_batches(d, B, S, T, n_img, 8, dev, 1234 + rank, ragged=a.lengths == "ragged")
    Bg = B * world
    from variational_mmt_amd.dp import GradSync
    sync = GradSync(eng)          # attaches itself to the engine when torch.distributed runs with > 1 rank

    eng.hold_back = True               # a training loop: every update is followed by a forward (TrainerMultimodal._train_loop sets the same)

    executed = {"steps": 0, "one_stream": 0}      # every step this process runs, timed or not: what a profile of the run is averaged over

    def step(i):
        executed["steps"] += 1
        executed["one_stream"] += 0 if eng.use_side_stream else 1
        src, sl, tgt, idx, tlen, n_tok = batches[i % len(batches)]
        ws = eng.forward(src, sl, tgt, idx, training=True, tgt_len=tlen if a.conditional else None, n_tgt_tokens=n_tok)
        eng.loss_backward(ws, normalization=Bg, batch_global=Bg)
        sync.all_reduce()          # waits for the segment all-reduces the backward plan issued behind each segment
        eng.optim_step(lr=0.002, max_grad_norm=5.0)
        return ws

    # ---- dominant-kernel timing hooks: events on the launch stream around the generator kernels -----------------
    ws0 = eng.workspace(B, S, Tp)        # (ragged lengths: the hooks sit on the full-length bucket only; other buckets run unhooked)
    dom = {"gen_fwd": [], "gen_bwd": []}

    def wrap(plan, index, key):
        fn, args, name, keep, sid = plan[index]

        def timed_fn(*x):
            if not dom.get("on"):
                return fn(*x)
            # events on the stream the kernel is launched on (the plan passes it as the last argument; the backward sweep of
            # the generator runs on the engine's side stream, which torch's current stream knows nothing about)
            st_ = torch.cuda.ExternalStream(x[-1]) if x[-1] else torch.cuda.current_stream()
            s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(st_)
            rc = fn(*x)
            e_.record(st_)
            dom[key].append((s, e_))
            if key == "gen_fwd" and name == "vmmt_gen_fwd_dO":
                dom.setdefault("gen_fwd_tokens", []).append(int(x[8]))     # (ragged batches: the sweep runs over the rows that carry a target)
            return rc
        timed_fn.__name__ = name
        plan[index] = (timed_fn, args, name, keep, sid)

    for j, (fn, args, name, keep, sid) in enumerate(ws0.plan_loss_train):
        if name in ("vmmt_gen_fwd_dO", "vmmt_gen_loss_fwd"):
            wrap(ws0.plan_loss_train, j, "gen_fwd")

    # warm-up: the W steps asked for, and at least one pass over the whole batch rotation, so that no input tensor (and no workspace
    # bucket of the ragged rotation) is seen for the first time inside the timed region
    n_warm = max(a.warmup, len(batches), 1)
    state0 = gpu_state()
    step(0)                            # builds the launch plans (the backward plan exists behind the first step)
    # backward plan exists now: hook the gen_loss_bwd entry
    for j, (fn, args, name, keep, sid) in enumerate(ws0.plan_bwd):
        if name in ("vmmt_gen_loss_bwd", "vmmt_gen_loss_bwd_db"):
            wrap(ws0.plan_bwd, j, "gen_bwd")

    def timed_region(first, hooks):
        """EXACTLY a.steps steps between barrier + synchronize on both sides (one perf_counter pair = the metric), plus what explains a
        slow region: one event per step on the main stream (device time of each step) and the host clock after each step's enqueue"""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dom["on"] = hooks
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        host = [0.0] * a.steps
        torch.cuda.synchronize()
        sch0 = host_sched()
        t0 = time.perf_counter()
        evs[0].record()
        ws_ = None
        for i in range(a.steps):
            ws_ = step(first + i)
            evs[i + 1].record()
            host[i] = time.perf_counter()
        t_enq = time.perf_counter()
        eng.wait_background()          # (a half of the last update that the engine held back for the next forward: inside the timed region)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        sch = sched_delta(sch0, host_sched())
        dom["on"] = False
        dev_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)]
        host_ms = [(host[i] - (host[i - 1] if i else t0)) * 1e3 for i in range(a.steps)]
        # the device's own span of the region: first event (recorded right behind t0) to the last step's event; what the region's
        # wall clock has beyond it is launch latency in front and the synchronize's return behind
        sch["device_span_ms"] = round(evs[0].elapsed_time(evs[a.steps]), 3)
        sch["region_ms"] = round(dt_ * 1e3, 3)
        return dt_, ws_, dev_ms, host_ms, (t_enq - t0) * 1e3, sch

    # self-check of the schedule on THIS box: the step is 3.4 ms of kernel time packed into ~1.7 ms by three streams of different priority
    # and persistent kernels that need their workgroups co-resident.  A few steps with everything on ONE stream against a few steps as
    # scheduled: a box (a driver, a queue configuration) on which the packing is lost shows a ratio near 1 and the line says so
    # the host's collector out of the timed regions: a full collection of a process that has imported torch walks ~10^6 objects (tens of
    # ms, the length of the whole 20-step region); nothing the steps allocate is cyclic, reference counting frees it.  In FRONT of the
    # self-check below, not behind it: the collection leaves the GPU idle for ~0.1 s, and a region that starts behind an idle gap ran its
    # first ten steps 3-10 % slow (clock ramp: step_ms lists of round 5's first leases)
    import gc
    gc.collect()
    gc.freeze()
    overlap = None
    if world == 1 and not a.no_side_stream and not a.no_overlap_probe:
        def timed_steps(n, first):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(n):
                step(first + i)
            eng.wait_background()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        eng.use_side_stream = False
        timed_steps(2, 0)
        serial = timed_steps(4, 2)
        eng.use_side_stream = True
        timed_steps(3, 0)
        packed = timed_steps(4, 3)
        overlap = {"one_stream_ms": round(serial, 3), "as_scheduled_ms": round(packed, 3), "ratio": round(serial / packed, 3),
                   "verdict": "streams overlap" if serial / packed > 1.12 else "NO OVERLAP: the side / aux streams' work does not run beside the main stream's on this box"}
    # the W warm-up steps LAST, right in front of the timed region
    for i in range(n_warm):
        step(i)
    regions = []
    n_rep = max(1, a.repeats)
    for r_ in range(n_rep):            # the FIRST region is the official one; the others show whether it was typical for the box
        regions.append(timed_region(n_warm + r_ * a.steps, hooks=(r_ == 0)) + (gpu_state(),))
    dt, ws, dev_ms, host_ms, enq_ms, sched = regions[0][:6]
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = eng.read_stats(ws, batch_global=B)
    eng.check_async_errors()           # settles a persistent-kernel hand-off that timed out (fallback + warning; reported below)
    dp_block = None
    if sync.active():
        # the multi-GPU line must be auditable: which backend ran, which form of each collective, how many ranks it saw, and what every
        # segment's collective cost -- timed over a few EXTRA steps behind the timed region (events on the step would perturb it)
        sync.timing, sync.exposed = [], []
        for i in range(min(10, max(3, a.steps))):
            sync.step_event = torch.cuda.Event(enable_timing=True)
            sync.step_event.record()
            step(a.warmup + a.steps + i)
        segs, exposed = sync.timing_report()
        sync.timing = None
        dp_block = {"backend": sync.backend, "world_seen": dist.get_world_size() if dist is not None else sync.world, "sharded_optimizer": bool(sync.sharded),
                    "native_collectives": sync.native_collectives(), "branches": ["%s: %s (%s)" % b_ for b_ in sync.branch_log],
                    "segments": segs, "exposed_ms": round(sum(exposed.values()), 4), "exposed_by_wait_ms": exposed,
                    "persistent_lstm": bool(eng.persistent_lstm)}

    if rank == 0:
        fl = flops_per_triplet(d, S, Tp)
        ms = dt / a.steps * 1e3
        value = Bg * a.steps / dt
        step_sum, slow_steps, step_med = step_summary(dev_ms, host_ms)
        reps = [r_[0] / a.steps * 1e3 for r_ in regions]
        diagnosis = diagnose(ms, a.steps, step_med, slow_steps, enq_ms, sched, reps, overlap)
        # dominant kernel = the fused vocabulary projection (+log-softmax+NLL) GEMM passes: 2*M*V*H FLOP per launch
        M = Tp * B
        fused = bool(getattr(ws0, "gen_fused", False))
        # per launch: one [M x V x H] product in the G^T path (generator.hip), two in the fused sweep (generator_fused.hip:
        # logits + dO; its time includes the combine kernel behind it)
        gen_flop = (4.0 if fused else 2.0) * M * d.vt * d.hid
        # per step: the backward pass may walk the vocabulary in several launches (Engine.gen_chunks); their times add up
        t_f = sum(s.elapsed_time(e_) for s, e_ in dom["gen_fwd"]) / max(1, a.steps)      # ms (fwd incl. combine)
        t_b = sum(s.elapsed_time(e_) for s, e_ in dom["gen_bwd"]) / max(1, a.steps)
        t_dom = max(t_f, t_b)
        ach = gen_flop / (t_dom * 1e-3) / 1e12 if t_dom > 0 else 0.0
        if fused and dom.get("gen_fwd_tokens"):      # per launch as issued: 4 H V x the launch's own token count over its own time
            tsum = sum(s.elapsed_time(e_) for s, e_ in dom["gen_fwd"])
            ach = 4.0 * sum(dom["gen_fwd_tokens"]) * d.vt * d.hid / (tsum * 1e-3) / 1e12 if tsum > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json).  The entry names the source
        # text it was measured on -- the kernel's own region of its .hip file (tools/traffic_key.py), so that an edit elsewhere in the
        # file does not void it, and tests/test_traffic_json.py fails while the region and the entry disagree
        traffic, traffic_note = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["config%s" % a.config]
            if B == 256 and a.dtype == "bf16" and not a.conditional and a.lengths == "fixed":
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import traffic_key
                traffic = tj["read_bytes"] + tj["write_bytes"]
                sha = traffic_key.key_hash(tj["kernel_region"])
                if sha != tj["kernel_region_sha16"]:
                    traffic_note = ("measured on an EARLIER revision of %s's source region (%s, now %s): re-run tools/pmc.sh" %
                                    (tj["kernel_region"], tj["kernel_region_sha16"], sha))
            else:
                traffic_note = "the PMC pass was taken at batch 256, bf16, fixed lengths, fixed prior: no figure for this variation"
        except Exception as ex:
            traffic_note = "no PMC figure for this configuration (%s)" % type(ex).__name__
        out = {
            "metric": "triplets/sec", "value": round(value, 1), "unit": "triplets/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (cf["name"] % B) + ", dropout %.1f, Adam%s%s" % (a.dropout, ", --conditional prior" if a.conditional else "",
                                                                                    ", lengths U[10,20]" if a.lengths == "ragged" else ""),
                       "global_batch": Bg, "parallelism": "dp%d" % world, "image_rows": n_img},
            "roofline": {"bound": "mfma", "kernel": (("gen2w_kernel" if d.hid > 512 else "gen2p_kernel" if d.hid > 256 else "gen2_kernel") + " (fused vocabulary sweep: logits + softmax statistics + dO, softmax weights stored for the dWg GEMM)" if fused else
                                    "gen_kernel (vocab projection + log-softmax/NLL pass, slower of fwd/bwd)"),
                         "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         # what the sweep must read at least -- Wg once, O once (bf16) -- against which `traffic` is to be judged: the rest of
                         # it is the softmax weights P [tokens][V] (bf16) + per-slice partials written for the dWg product, which reads P back
                         "traffic_operands": (d.vt * d.hid + M * d.hid) * 2 if fused else None,
                         "traffic_P_written": (M * d.vt * 2) if fused else None,
                         "ms_fwd": round(t_f, 4), "ms_bwd": round(t_b, 4),
                         "step_tflops": round(3 * fl["total"] * Bg / (dt / a.steps) / 1e12, 2)},
            # what explains the value: device time of every step of the official region (events on the main stream), the host's time to
            # enqueue it, the same region repeated, the steps that took > 1.5 x the median, the box's clocks before / after
            "step_ms": step_sum, "value_median": round(Bg / (step_med * 1e-3), 1),
            "host_enqueue_ms": {"total": round(enq_ms, 3), "per_step_median": round(sorted(host_ms)[len(host_ms) // 2], 4),
                                "per_step_max": round(max(host_ms), 4)},
            "repeats_ms": [round(r_[0] / a.steps * 1e3, 4) for r_ in regions],
            "repeats_step_median_ms": [step_summary(r_[2], r_[3])[0]["median"] for r_ in regions],
            "slow_steps": slow_steps, "diagnosis": diagnosis, "warmup_run": n_warm, "stream_overlap": overlap,
            # every step this process has run up to here (plan-building step, overlap probe, warm-up, all timed regions): the divisor of a
            # per-step kernel profile of this very command (tools/prof.sh reads it; a once-per-step kernel must then show 1.0 calls / step)
            "steps_executed": dict(executed),
            "host_sched": dict(sched, cpu_quota=cpu_quota(), repeats=[r_[5] for r_ in regions[1:]]),
            "gpu_state": {"before_warmup": state0, "after_each_region": [r_[6] for r_ in regions]},
            "elbo_per_sentence": round(st["elbo"] / B, 4),
            "seq_fallbacks": eng.seq_fallbacks, "steps_skipped": eng.steps_skipped,
            # how the engine scheduled the optimiser step in this run: the side-stream half of an update held back until the next forward's
            # head is through (two or more layers, one rank) or issued at once; the last one is flushed inside the timed region
            "held_back_update": bool(eng.hold_back and eng.bg_after_head and not a.conditional and not sync.active()),
        }
        if dp_block is not None:
            out["dp"] = dp_block
        if world == 1 and not a.no_parity:
            try:
                out["parity"] = parity_gate(dev, a.dtype, a.dropout)
            except Exception as ex:          # the gate failing to RUN must not take the measured line with it; it is reported as such
                out["parity"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
