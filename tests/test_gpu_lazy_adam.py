"""GPU: the row-wise (lazy) Adam of the embedding tables (csrc/optim.hip: vmmt_rows_mark / vmmt_adam_rows_catchup / vmmt_adam_rows_step /
vmmt_gather_rows_lazy / vmmt_sumsq_rows; engine._build_lazy) against dense Adam.

Reference: torch.optim.Adam over every element at every step (onmt/Optim.py:68-70,94-96).  A row without gradient still moves under
dense Adam (its moments decay, the parameter follows them); the lazy path replays exactly those zero-gradient steps when the row is next
used, so parameters AND moments must come out BIT-identical -- checked at the kernel level, where both paths can be fed identical
gradients (the step's own gradients carry float-atomic noise from run to run)."""
import ctypes as C

import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu


def test_rows_kernels_bit_identical_to_dense_adam():
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    R, Cc, steps = 301, 500, 60
    g = torch.Generator().manual_seed(3)
    p0 = (torch.rand(R, Cc, generator=g) - 0.5).to(dev)
    dense = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0))
    lazy = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0))
    grad_d, grad_l = torch.zeros_like(p0), torch.zeros_like(p0)
    flags, last = torch.zeros(R, dtype=torch.int32, device=dev), torch.zeros(R, dtype=torch.int32, device=dev)
    rowsq = torch.zeros(R, dtype=torch.float32, device=dev)
    hist = torch.zeros(2 * (steps + 2), dtype=torch.float32, device=dev)
    sq_d, sq_l = (torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=dev) for _ in range(2))
    b1, b2, eps = 0.9, 0.999, 1e-9

    def same(tag):
        torch.cuda.synchronize()
        for k in ("p", "m", "v"):
            assert torch.equal(dense[k], lazy[k]), (tag, k, (dense[k] - lazy[k]).abs().max().item())

    for step in range(1, steps + 1):
        lr = 0.002 if step < 25 else 0.001                      # a learning-rate decay in the middle (replays read it from `hist`)
        max_norm = 5.0 if step % 3 else 0.02                    # clipping active on every third step
        n_ids = int(torch.randint(1, 40, (1,), generator=g))
        ids = torch.randint(0, R, (n_ids,), generator=g)
        if step % 7 == 0:
            ids = torch.cat([ids, ids[:3]])                     # duplicates in a batch
        rows = torch.unique(ids)
        ids_d = ids.to(dev)
        # lazy: flag + catch up + clear the flagged rows' gradient, then "backward" writes this step's gradient
        L.check(lib.vmmt_rows_mark(ids_d.data_ptr(), ids_d.numel(), flags.data_ptr(), R, st), "mark")
        L.check(lib.vmmt_adam_rows_catchup(lazy["p"].data_ptr(), grad_l.data_ptr(), lazy["m"].data_ptr(), lazy["v"].data_ptr(), R, Cc,
                                           flags.data_ptr(), last.data_ptr(), hist.data_ptr(), b1, b2, eps, step - 1, 1, st), "catchup")
        torch.cuda.synchronize()
        assert (grad_l[flags.bool()] == 0).all()
        # rows about to be gathered are current: equal to the dense parameters right now
        assert torch.equal(lazy["p"][rows.to(dev)], dense["p"][rows.to(dev)]), step
        # ... and the lazy lookup of ANY rows (stale ones included) returns the dense parameters without touching the table
        probe = torch.randint(0, R, (37,), generator=g).to(dev)
        outp = torch.zeros(37, Cc, dtype=torch.float32, device=dev)
        before = lazy["p"].clone()
        L.check(lib.vmmt_gather_rows_lazy(L.F32, lazy["p"].data_ptr(), lazy["m"].data_ptr(), lazy["v"].data_ptr(), Cc, probe.data_ptr(),
                                          outp.data_ptr(), Cc, 37, last.data_ptr(), hist.data_ptr(), b1, b2, eps, step - 1, st), "gather_lazy")
        torch.cuda.synchronize()
        assert torch.equal(outp, dense["p"][probe]) and torch.equal(before, lazy["p"]), step
        gr = (torch.rand(rows.numel(), Cc, generator=g) - 0.5).to(dev) * (10.0 if step % 5 == 0 else 0.1)
        grad_d.zero_()
        grad_d[rows.to(dev)] = gr
        grad_l[rows.to(dev)] = gr
        # norms: dense over the whole table, lazy over the flagged rows; equal up to summation order
        sq_d[:L.SUMSQ_SLOTS].zero_()
        sq_l[:L.SUMSQ_SLOTS].zero_()
        L.check(lib.vmmt_sumsq(grad_d.data_ptr(), R * Cc, sq_d.data_ptr(), 0, st), "sumsq")
        L.check(lib.vmmt_sumsq_rows(grad_l.data_ptr(), R, Cc, flags.data_ptr(), rowsq.data_ptr(), sq_l.data_ptr(), 3, st), "sumsq_rows")
        torch.cuda.synchronize()
        a, b = float(sq_d[0]), float(sq_l[3])
        assert abs(a - b) <= 2e-6 * a, (step, a, b)
        # the same clip coefficient for both (the bit-level claim is about the update, not about the norm's summation order)
        L.check(lib.vmmt_adam_step(dense["p"].data_ptr(), grad_d.data_ptr(), dense["m"].data_ptr(), dense["v"].data_ptr(), R * Cc, lr, b1, b2,
                                   eps, step, max_norm, sq_d.data_ptr(), 1.0, 0, None, st), "adam")
        L.check(lib.vmmt_adam_rows_step(lazy["p"].data_ptr(), grad_l.data_ptr(), lazy["m"].data_ptr(), lazy["v"].data_ptr(), R, Cc,
                                        flags.data_ptr(), last.data_ptr(), hist.data_ptr(), lr, b1, b2, eps, step, max_norm, sq_d.data_ptr(),
                                        1.0, st), "rows_step")
        torch.cuda.synchronize()
        assert int(flags.sum()) == 0 and (last[rows.to(dev)] == step).all()
        if step in (1, 17, 40, steps):                              # flush everything and compare the whole table
            L.check(lib.vmmt_adam_rows_catchup(lazy["p"].data_ptr(), None, lazy["m"].data_ptr(), lazy["v"].data_ptr(), R, Cc, None,
                                               last.data_ptr(), hist.data_ptr(), b1, b2, eps, step, 0, st), "flush")
            same(step)
            assert (last == step).all()
    # some row was never touched after an early update and still moved (the decaying moments): the flush is not a no-op
    assert (dense["p"] != p0).any(dim=1).float().mean().item() > 0.9


def _engine(c, p, lazy, dtype="f32"):
    from variational_mmt_amd.engine import Dims, Engine
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype=dtype, device="cuda", seed=1)
    if not lazy:
        e.lazy_rows = False
    assert e.lazy_active() == lazy
    e.load_state_dict(p)
    return e


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_engine_lazy_rows_equal_dense_over_many_steps(dtype):
    """the training step with the lazy tables against the same step with dense Adam: 14 updates on changing batches (every batch touches
    other rows), an evaluation pass in between, learning-rate change, state_dict in the middle; parameters and moments agree to the
    run-to-run noise of the step itself (float atomics in the gradient products), and rows no batch ever used are EXACTLY the initial
    values moved by nothing (their gradient never was anything but zero)"""
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True)
    p = O.init_params(c, seed=2)
    engines = {lz: _engine(c, p, lz, dtype) for lz in (True, False)}
    used_src, used_tgt = set(), set()
    for step in range(14):
        bt = O.synth_batch(c, 6, 5 + step % 3, 6 + step % 2, n_img=12, seed=50 + step, fixed_len=False)
        used_src |= set(bt["src"].reshape(-1).tolist())
        used_tgt |= set(bt["tgt"][:-1].reshape(-1).tolist())
        for lz, e in engines.items():
            e.set_image_table(bt["table"])
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
            e.loss_backward(ws, normalization=6)
            e.optim_step(lr=0.01 if step < 8 else 0.004, max_grad_norm=5.0 if step % 4 else 0.5)
            if step == 5:        # an evaluation pass between updates (validation inside an epoch)
                ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=False)
                e.loss(ws)
            if step == 9:
                sd = e.state_dict()                      # flushes
                assert all(torch.isfinite(v).all() for v in sd.values())
    torch.cuda.synchronize()
    a, b = engines[True], engines[False]
    assert a.lazy_tables and a._lazy_dirty
    a.flush_lazy_rows()
    torch.cuda.synchronize()
    tol = 5e-6 if dtype == "f32" else 2e-3
    for x, y, what in ((a.flat_p, b.flat_p, "p"), (a.flat_m, b.flat_m, "m"), (a.flat_v, b.flat_v, "v")):
        n = a.n_opt
        err = (x[:n] - y[:n]).abs().max().item()
        assert err <= tol * max(1.0, y[:n].abs().max().item()), (what, err)
    for name, used in (("encoder.embeddings.make_embedding.emb_luts.0.weight", used_src), ("decoder.embeddings.make_embedding.emb_luts.0.weight", used_tgt)):
        idle = [r for r in range(p[name].shape[0]) if r not in used]
        assert idle
        for e in (a, b):
            assert torch.equal(e.params[name][idle].cpu(), p[name][idle]), name
    # the lazy engine zeroes / norms / updates only the flagged rows: its plans carry the row-wise entries
    ws = a.workspace(6, 5, 5)
    names = [en[2] for en in ws.plan_fwd_train]
    assert names.count("vmmt_rows_mark") == 2 and names.count("vmmt_gather_rows_lazy") == 2 and names.count("vmmt_gather_rows") == 1
