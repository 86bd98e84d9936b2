"""RCCL executes.  The test box has ONE GPU, and RCCL refuses two ranks on one device -- so every multi-rank test of this repository
runs on gloo, whose lack of the tensor collectives sends dp.GradSync down its fallback branch (all-reduce / per-owner broadcasts).
Here a process group of world size ONE is created on the `nccl` backend (= RCCL on ROCm) and GradSync is forced to attach to it
(VMMT_DP_FORCE=1): every collective of the data-parallel step -- the in-place reduce_scatter_tensor of each arena segment from inside
the backward plan, the KL all-reduce, the norm all-gather, the in-place all_gather_into_tensor of the parameters, gather_moments,
broadcast_replica -- runs through RCCL's kernels with itself as the only peer, next to the persistent recurrence kernels, on the
engine's own streams.  With one rank every collective is the identity, so the step must equal the step without data parallelism.
Runs in a child process: the process group must not leak into the other tests."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", VMMT_DP_FORCE="1", VMMT_DP_NATIVE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from oracle import vi1_oracle as O
    from variational_mmt_amd.dp import GradSync
    from variational_mmt_amd.engine import Dims, Engine
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    c = O.Cfg(vs=61, vt=300, emb=64, hid=256, z=128, layers=1, brnn=True)        # the persistent recurrences + fused sweep serve it
    p = O.init_params(c, seed=1)
    B = 32
    bts = [O.synth_batch(c, B=B, S=7, T=8, n_img=40, seed=80 + i, fixed_len=False) for i in range(3)]
    res = {}
    finals = {}
    for mode in ("plain", "sharded", "replicated", "sharded_c10d"):
        e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="bf16", device=dev, seed=2)
        e.load_state_dict(p)
        e.set_image_table(bts[0]["table"])
        sync = None
        if mode != "plain":
            # (direct=False: torch.distributed's own tensor collectives instead of the direct RCCL calls -- the path a failed self-check leaves)
            sync = GradSync(e, sharded=(mode != "replicated"), direct=(mode != "sharded_c10d"))
            assert e.dp is sync and sync.active() and sync.world == 1 and sync.backend == "nccl"
        for i in range(3):
            bt = bts[i]
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
            e.loss_backward(ws, normalization=B, batch_global=B, use_freebits=True, margin=0.01)        # (free bits: the KL all-reduce matters)
            if sync is not None:
                sync.all_reduce()
            e.optim_step(lr=0.002, max_grad_norm=0.5)
        torch.cuda.synchronize()
        names = [en[2] for en in ws.plan_fwd_train] + [en[2] for en in ws.plan_bwd]
        res[mode] = dict(persistent="vmmt_lstm_seq_fwd" in names and "vmmt_lstm_seq_bwd" in names, errors=e.lstm_seq_errors(),
                         collectives=[n for n in names if n in ("ALLREDUCE", "KL_ALLREDUCE")], fused=bool(ws.gen_fused))
        if sync is not None:
            res[mode]["native"] = sync.native_collectives()
            res[mode]["branches"] = list(sync.branch_log)
            res[mode]["kl_global"] = float(ws.kl_global.item())
            res[mode]["kl_local"] = float(ws.stats[3].item())
            sync.gather_moments()
            sync.broadcast_replica(0)
            res[mode]["identical"] = sync.replicas_identical()
            torch.cuda.synchronize()
        e.check_async_errors()
        res[mode]["fallbacks"] = e.seq_fallbacks
        finals[mode] = (e.flat_p[:e.n_opt].cpu().clone(), e.flat_m[:e.n_opt].cpu().clone())
    for mode in ("sharded", "replicated", "sharded_c10d"):
        res[mode]["p_err"] = float((finals[mode][0] - finals["plain"][0]).abs().max())
        res[mode]["m_rel"] = float((finals[mode][1] - finals["plain"][1]).norm() / finals["plain"][1].norm())
    torch.save(res, out)
    dist.destroy_process_group()


def test_world_one_nccl_group_runs_every_collective_natively(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 26000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, out), nprocs=1, join=True)
    r = torch.load(out)
    for mode in ("plain", "sharded", "replicated", "sharded_c10d"):
        assert r[mode]["persistent"] and r[mode]["fused"] and not any(r[mode]["errors"]) and r[mode]["fallbacks"] == 0, (mode, r[mode])
    # the backward plan issued one gradient collective per arena segment + the KL all-reduce
    assert r["sharded"]["collectives"].count("ALLREDUCE") == 4 and "KL_ALLREDUCE" in r["sharded"]["collectives"]
    # the branch RCCL takes: the tensor collectives, in place -- not gloo's fallback
    assert r["sharded"]["native"] == {"direct_rccl": True}, r["sharded"]            # the step's collectives straight through RCCL's C API
    assert r["sharded_c10d"]["native"] == {"reduce_scatter": True, "all_gather": True, "direct_rccl": False}, r["sharded_c10d"]
    assert ("step collectives", "torch.distributed") in [b[:2] for b in r["sharded_c10d"]["branches"]]      # (direct=False: said so in the branch log)
    assert all(b[1] == "native" for b in r["sharded_c10d"]["branches"] if b[0] != "step collectives")
    for mode in ("sharded", "replicated", "sharded_c10d"):
        x = r[mode]
        assert x["identical"] and abs(x["kl_global"] - x["kl_local"]) <= 1e-6 * abs(x["kl_local"]), (mode, x)
        # one rank: every collective is the identity -> the same three updates as without data parallelism (float atomics apart)
        assert x["p_err"] <= 2e-3 and x["m_rel"] <= 1e-3, (mode, x)
