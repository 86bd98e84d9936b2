"""CPU, world_size 2 over gloo: the data-parallel convention of variational_mmt_amd.dp (normalise by the GLOBAL batch, KL
batch mean over the GLOBAL batch, SUM the ranks' gradients) reproduces the single-process step on the concatenated batch.
The per-rank compute here is the CPU oracle (test infrastructure); the code under test is dp.GradSync and the convention."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import vi1_oracle as O
    from variational_mmt_amd.dp import GradSync
    from variational_mmt_amd.engine import Dims
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = O.Cfg(vs=31, vt=37, emb=10, hid=12, z=6, img=2048, layers=1, brnn=True)
    p = O.init_params(c, seed=2)
    Bg = 6
    bt = O.synth_batch(c, Bg, 5, 6, n_img=8, seed=3, fixed_len=False)
    img = bt["table"][bt["indices"]]
    # a length-sorted global batch split in contiguous halves keeps every half sorted (packed-sequence requirement)
    sl = slice(rank * Bg // world, (rank + 1) * Bg // world)
    _, _, g = O.step_grads(p, c, bt["src"][:, sl], bt["src_len"][sl], bt["tgt"][:, sl], img[sl], bt["eps"][sl],
                           normalization=Bg, batch_global=Bg)
    # flat arena in the engine's order (gradient-carrying parameters only)
    d = Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn)
    names = [n for n, _ in d.param_shapes()[0]]
    flat = torch.cat([g[n].reshape(-1) for n in names])
    sync = GradSync(flat=flat, bucket_elems=100000)
    assert sync.world == world and len(sync.buckets()) > 1
    assert sync.global_batch(Bg // world) == Bg
    sync.all_reduce()
    if rank == 0:
        _, _, gfull = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"])
        want = torch.cat([gfull[n].reshape(-1) for n in names])
        err = (flat - want).abs().max().item() / want.abs().max().item()
        torch.save({"err": err}, out)
    dist.destroy_process_group()


def test_dp_sum_of_rank_gradients_equals_single_process(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    err = torch.load(out)["err"]
    assert err < 2e-5, err
