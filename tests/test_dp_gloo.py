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


def _worker_freebits(rank, world, port, out):
    """free bits under data parallelism (VILoss.py:463-476): max(m * KL_mean_GLOBAL, margin) -- the ranks exchange ONE float (their
    KL sums) before the KL term is differentiated; checked on both sides of the margin"""
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
    sl = slice(rank * Bg // world, (rank + 1) * Bg // world)
    d = Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn)
    names = [n for n, _ in d.param_shapes()[0]]

    def reduce_kl(local_sum):
        t = local_sum.detach().clone().reshape(1)
        dist.all_reduce(t)
        return float(t)

    _, L0, _ = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"])
    kl_mean = float(L0["kl_before"])
    res = {}
    for tag, margin in (("below", kl_mean * 0.5), ("above", kl_mean * 2.0)):     # KL above the margin / clamped to the margin
        _, Ll, g = O.step_grads(p, c, bt["src"][:, sl], bt["src_len"][sl], bt["tgt"][:, sl], img[sl], bt["eps"][sl], normalization=Bg,
                                batch_global=Bg, use_freebits=True, freebits=margin, kl_global_sum=reduce_kl)
        flat = torch.cat([g[n].reshape(-1) if n in g else torch.zeros(int(torch.tensor(shp).prod())) for n, shp in d.param_shapes()[0]])
        sync = GradSync(flat=flat, bucket_elems=100000)
        sync.all_reduce()
        elbo = torch.tensor([float(Ll["elbo"])], dtype=torch.float64)
        dist.all_reduce(elbo)                       # the ranks' shares add up to the single-process ELBO
        if rank == 0:
            _, Lf, gf = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], use_freebits=True, freebits=margin)
            want = torch.cat([gf[n].reshape(-1) if n in gf else torch.zeros(int(torch.tensor(shp).prod())) for n, shp in d.param_shapes()[0]])
            res[tag] = ((flat - want).abs().max().item() / want.abs().max().item(), abs(float(elbo) - float(Lf["elbo"])) / abs(float(Lf["elbo"])),
                        float((want[-100:] ** 2).sum()))
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


def test_dp_freebits_uses_the_global_kl_mean(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_worker_freebits, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    for tag in ("below", "above"):
        assert res[tag][0] < 2e-5 and res[tag][1] < 1e-6, (tag, res[tag])


def test_sharded_iterator_partitions_every_global_minibatch():
    """onmt.io.OrderedIterator(dp_rank=, dp_world=): the ranks walk the same global minibatches and take disjoint, sorted,
    equally sized (+-1) shares; the global batch size / token count ride on the batch (no collective in the trainer)."""
    from variational_mmt_amd.onmt import io
    G = os.path.join(ROOT, "tests", "golden", "textdata", "demo")
    ds = io.load_dataset(G + ".train.1.pt")
    fields = io.load_fields_from_vocab(io.load_vocab(G + ".vocab.pt"), "text")
    ds.fields = dict((k, f) for k, f in fields.items() if k in ds.examples[0].__dict__)
    world, bs = 3, 4
    its = [io.OrderedIterator(dataset=ds, batch_size=bs, device=None, sort=False, train=True, sort_within_batch=True, repeat=False,
                              dp_rank=r, dp_world=world, dp_seed=7) for r in range(world)]
    per_rank = [list(it) for it in its]
    assert len({len(b) for b in per_rank}) == 1                  # every rank sees the same number of batches
    seen = []
    for step in zip(*per_rank):
        gb = {b.global_batch_size for b in step}
        assert len(gb) == 1 and sum(b.batch_size for b in step) == gb.pop()
        assert len({b.global_ntokens for b in step}) == 1
        assert step[0].global_ntokens == sum(int(b.tgt[0][1:].ne(1).sum()) for b in step)
        sizes = [b.batch_size for b in step]
        assert max(sizes) - min(sizes) <= 1
        for b in step:
            lens = b.src[1].tolist()
            assert lens == sorted(lens, reverse=True)            # each share is still sorted by decreasing source length
            seen += b.indices.tolist()
    assert len(seen) == len(set(seen))                           # disjoint
    assert len(ds) - len(seen) < world                           # at most world-1 examples (a tail minibatch) are skipped


def _worker_shards(rank, world, port, out):
    """dp.GradSync's sharded-optimiser collectives on plain tensors: ownership arithmetic, reduce-scatter (own shard = the sum),
    all-gather (every shard back on every rank), for segments that split evenly and for one with an uneven tail"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from variational_mmt_amd.dp import GradSync
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sync = GradSync(flat=torch.zeros(1), sharded=True)
    assert sync.world == world and sync.sharded
    segs = [(0, 512 * 3), (512 * 3, 512 * 3 + 512 * 40), (512 * 43, 512 * 43 + 64 * 13)]      # the last one: 13 units over `world` ranks
    n = segs[-1][1]
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.rand(n, generator=g)
    everyone = [torch.rand(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    want = torch.stack(everyone).sum(0)
    flat = mine.clone()
    ok = True
    covered = torch.zeros(n, dtype=torch.int64)
    for lo, hi in segs:
        # ownership: the ranks' shards tile the segment exactly once, whole 64-element units
        for r in range(world):
            a, b = GradSync.shard(type("R", (), dict(world=world, rank=r))(), lo, hi)
            assert lo <= a <= b <= hi and (a - lo) % 64 == 0 and ((b - lo) % 64 == 0 or b == hi)
            if rank == 0:
                covered[a:b] += 1
        sync.reduce_scatter(flat, lo, hi).wait()
        a, b = sync.shard(lo, hi)
        ok = ok and bool(torch.allclose(flat[a:b], want[a:b], rtol=1e-6, atol=0))
    if rank == 0:
        assert (covered == 1).all()
    # "Adam" on the own shards: an element-wise function of the reduced gradient, then the parameters travel back
    p = torch.zeros(n)
    for lo, hi in segs:
        a, b = sync.shard(lo, hi)
        p[a:b] = flat[a:b] * 0.5 + 1.0
    for lo, hi in segs:
        sync.all_gather(p, lo, hi).wait()
    ok = ok and bool(torch.allclose(p, want * 0.5 + 1.0, rtol=1e-6, atol=0))
    rows = sync.all_gather_rows(torch.tensor([float(rank), 2.0 * rank]))
    ok = ok and rows.shape == (world, 2) and rows[:, 0].tolist() == [float(r) for r in range(world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, (ok, p.double().sum().item()))
    if rank == 0:
        torch.save({"ok": all(x[0] for x in gathered), "same": len({x[1] for x in gathered}) == 1}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_optimiser_collectives(tmp_path, world):
    out = str(tmp_path / "res.pt")
    port = 27000 + (os.getpid() % 2000) + world
    mp.spawn(_worker_shards, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out)
    assert r["ok"] and r["same"], r


def _worker_probe(rank, world, port, out):
    """dp.GradSync pre-flight behaviour (VERDICT r3, weak #6a/b): the probe says which form of a collective it chose, a backend that
    lacks the tensor form falls back ONLY for "not supported", VMMT_DP_NATIVE=1 turns that into an error, and an engine that cannot be
    attached stops the run instead of leaving unsynchronised replicas behind"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from variational_mmt_amd.dp import GradSync
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    sync = GradSync(flat=torch.zeros(1), sharded=True)
    t = torch.zeros(512 * world)
    res["rs_native"] = sync._probe("reduce_scatter", t)
    res["log"] = list(sync.branch_log)
    res["backend"] = sync.backend
    os.environ["VMMT_DP_NATIVE"] = "1"
    strict = GradSync(flat=torch.zeros(1), sharded=True)
    try:
        strict._probe("reduce_scatter", t)
        res["strict_raised"] = False
    except RuntimeError as ex:
        res["strict_raised"] = "probe failed" in str(ex)
    os.environ["VMMT_DP_NATIVE"] = "0"
    off = GradSync(flat=torch.zeros(1), sharded=True)
    res["forced_fallback"] = (off._probe("all_gather", t) is False) and off.branch_log[-1][1] == "fallback"
    del os.environ["VMMT_DP_NATIVE"]

    class _Broken(object):
        dp = None
        rng_counter = 0
        row_tables = ()

        def flush_lazy_rows(self):
            pass

        def drop_workspaces(self):
            raise RuntimeError("boom")
    eng = _Broken()
    try:
        GradSync(eng)
        res["attach_raised"] = False
    except RuntimeError:
        res["attach_raised"] = eng.dp is None

    class _Dense(_Broken):
        dense_optimizer = True

        def drop_workspaces(self):
            pass
    d = _Dense()
    res["dense_not_sharded"] = (GradSync(d).sharded is False) and d.dp is not None
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_probe_reports_its_branch_and_attach_errors_are_fatal(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 28200 + (os.getpid() % 700)
    mp.spawn(_worker_probe, args=(2, port, out), nprocs=2, join=True)
    for r in torch.load(out):
        # (this torch's gloo has reduce_scatter_tensor for CPU tensors; for CUDA tensors -- the GPU tests -- it does not)
        assert r["backend"] == "gloo" and r["rs_native"] in (True, False), r
        assert r["log"] and r["log"][0][0] == "reduce_scatter" and r["log"][0][1] == ("native" if r["rs_native"] else "fallback"), r
        assert r["strict_raised"] == (not r["rs_native"]), r          # VMMT_DP_NATIVE=1: no silent fallback
        assert r["forced_fallback"] and r["attach_raised"] and r["dense_not_sharded"], r


class _FakeRccl(object):
    """stand-in for librccl in the bring-up protocol of dp._Rccl: the calls the stages make, each able to fail on chosen ranks"""
    def __init__(self, rank, fail_uid=False, fail_init=False):
        import ctypes as C
        self.rank, self.fail_uid, self.fail_init, self.destroyed = rank, fail_uid, fail_init, 0

        class _Fn(object):
            def __init__(s, f):
                s.f = f

            def __call__(s, *a):
                return s.f(*a)
        self.ncclGetUniqueId = _Fn(lambda p: 1 if self.fail_uid else (C.memmove(p, b"\x07" * 128, 128) and 0))
        self.ncclCommInitRank = _Fn(lambda pc, world, uid, rank: 5 if self.fail_init else self._init(pc, uid))
        self.ncclCommDestroy = _Fn(self._destroy)
        self.ncclGetErrorString = _Fn(lambda rc: b"stand-in error %d" % rc)
        self.ncclReduceScatter = self.ncclAllGather = self.ncclAllReduce = _Fn(lambda *a: 0)

    def _init(self, pc, uid):
        import ctypes as C
        assert bytes(uid.internal)[:1] in (b"\x07", b"")          # the id rank 0 drew reached this rank
        C.cast(pc, C.POINTER(C.c_void_p))[0] = 0x1234
        return 0

    def _destroy(self, comm):
        self.destroyed += 1
        return 0


def _worker_bringup(rank, world, port, out):
    """whatever fails during the direct communicator's bring-up, on whichever rank, EVERY rank leaves it the same way
    (an exception -> the run stays on torch.distributed) and none is left alone in a blocking call"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from variational_mmt_amd.dp import _Rccl
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    cases = {"fine": {}, "uid_fails_on_rank0": dict(fail_uid=(rank == 0)), "init_fails_on_rank1": dict(fail_init=(rank == 1)),
             "library_missing_on_rank1": "missing" if rank == 1 else {}}
    for name, kw in cases.items():
        lib = kw if kw == "missing" else _FakeRccl(rank, **kw)
        try:
            r = _Rccl(dist, torch.device("cpu"), lib=lib)
            res[name] = ("up", r.comm.value if r.comm else None)
            r.close()
            res[name] += (lib.destroyed,)
        except RuntimeError as ex:
            res[name] = ("raised", "this rank" in str(ex), getattr(lib, "destroyed", None))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_direct_rccl_bringup_fails_on_every_rank_or_on_none(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 26100 + (os.getpid() % 700)
    mp.spawn(_worker_bringup, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out)
    assert r0["fine"] == ("up", 0x1234, 1) and r1["fine"] == ("up", 0x1234, 1)            # communicator created, destroyed by close()
    for case, culprit in (("uid_fails_on_rank0", 0), ("init_fails_on_rank1", 1), ("library_missing_on_rank1", 1)):
        for rank, r in enumerate((r0, r1)):
            assert r[case][0] == "raised" and r[case][1] == (rank == culprit), (case, rank, r[case])
    assert r0["init_fails_on_rank1"][2] == 1        # the rank whose communicator did come up destroys it again


def test_direct_rccl_is_opt_in(monkeypatch):
    """until a multi-GPU parity run has passed, a data-parallel job uses torch.distributed's collectives unless it asks for the direct calls"""
    import inspect
    from variational_mmt_amd import dp
    src = inspect.getsource(dp.GradSync.__init__)
    assert 'os.environ.get("VMMT_DP_DIRECT", "0") == "1"' in src
