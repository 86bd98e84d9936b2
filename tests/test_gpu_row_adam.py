"""GPU: row-wise gradient bookkeeping of the embedding tables (csrc/optim.hip: vmmt_rows_mark / vmmt_rows_zero / vmmt_sumsq_rows /
vmmt_adam_rows_step; engine._build_row_tables) against the dense path.

Reference: loss.backward() into zero-filled .grad tensors, clip_grad_norm and torch.optim.Adam over EVERY element at every step
(onmt/TrainerMultimodal.py:628-629, onmt/Optim.py:68-70,94-96).  The row-wise path clears / norms / reads the gradient of the batch's
rows only and still updates every row: parameters and moments must be BIT-identical -- checked at the kernel level, where both paths can
be fed identical gradients (the step's own gradients carry float-atomic noise from run to run)."""
import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu


def test_rows_kernels_bit_identical_to_dense_adam():
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    R, Cc, steps = 301, 500, 40
    g = torch.Generator().manual_seed(3)
    p0 = (torch.rand(R, Cc, generator=g) - 0.5).to(dev)
    dense = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0))
    rows_ = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0))
    grad_d, grad_r = torch.zeros_like(p0), torch.zeros_like(p0)
    flags = torch.zeros(R, dtype=torch.int32, device=dev)
    rowsq = torch.zeros(R, dtype=torch.float32, device=dev)
    sq_d, sq_r = (torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=dev) for _ in range(2))
    b1, b2, eps = 0.9, 0.999, 1e-9
    for step in range(1, steps + 1):
        lr = 0.002 if step < 25 else 0.001
        max_norm = 5.0 if step % 3 else 0.02                    # clipping active on every third step
        ids = torch.randint(0, R, (int(torch.randint(1, 40, (1,), generator=g)),), generator=g)
        if step % 7 == 0:
            ids = torch.cat([ids, ids[:3]])                     # duplicates in a batch
        rows = torch.unique(ids).to(dev)
        ids_d = ids.to(dev)
        L.check(lib.vmmt_rows_mark(ids_d.data_ptr(), ids_d.numel(), flags.data_ptr(), R, st), "mark")
        L.check(lib.vmmt_rows_zero(grad_r.data_ptr(), R, Cc, flags.data_ptr(), st), "zero")
        torch.cuda.synchronize()
        assert int(flags.sum()) == rows.numel() and (grad_r == 0).all()         # last step's rows were cleared by the flags of THIS step or hold zeros
        gr = (torch.rand(rows.numel(), Cc, generator=g) - 0.5).to(dev) * (10.0 if step % 5 == 0 else 0.1)
        grad_d.zero_()
        grad_d[rows] = gr
        grad_r[rows] = gr
        sq_d[:L.SUMSQ_SLOTS].zero_()
        sq_r[:L.SUMSQ_SLOTS].zero_()
        L.check(lib.vmmt_sumsq(grad_d.data_ptr(), R * Cc, sq_d.data_ptr(), 0, st), "sumsq")
        L.check(lib.vmmt_sumsq_rows(grad_r.data_ptr(), R, Cc, flags.data_ptr(), rowsq.data_ptr(), sq_r.data_ptr(), 3, st), "sumsq_rows")
        torch.cuda.synchronize()
        a, b = float(sq_d[0]), float(sq_r[3])
        assert abs(a - b) <= 2e-6 * a, (step, a, b)             # the same norm up to the order of summation
        # the same clip coefficient for both (the bit-level claim is about the update, not about the norm's summation order)
        L.check(lib.vmmt_adam_step(dense["p"].data_ptr(), grad_d.data_ptr(), dense["m"].data_ptr(), dense["v"].data_ptr(), R * Cc, lr, b1, b2,
                                   eps, step, max_norm, sq_d.data_ptr(), 1.0, 0, None, None, st), "adam")
        L.check(lib.vmmt_adam_rows_step(rows_["p"].data_ptr(), grad_r.data_ptr(), rows_["m"].data_ptr(), rows_["v"].data_ptr(), R, Cc,
                                        flags.data_ptr(), lr, b1, b2, eps, step, max_norm, sq_d.data_ptr(), 1.0, None, st), "rows_step")
        torch.cuda.synchronize()
        assert int(flags.sum()) == 0
        for k in ("p", "m", "v"):
            assert torch.equal(dense[k], rows_[k]), (step, k, (dense[k] - rows_[k]).abs().max().item())
        # the gradient rows of this step stay in place until the next batch's flags clear them: leave garbage in an UNFLAGGED row to
        # show that the update never reads it
        grad_r[rows] = 0
        if step % 4 == 0:
            idle = int(torch.randint(0, R, (1,), generator=g))
            grad_r[idle] = 123.0
            grad_d[idle] = 0.0
            L.check(lib.vmmt_rows_mark(torch.tensor([idle], device=dev).data_ptr(), 1, flags.data_ptr(), R, st), "mark")
            L.check(lib.vmmt_rows_zero(grad_r.data_ptr(), R, Cc, flags.data_ptr(), st), "zero")
            torch.cuda.synchronize()
            assert (grad_r[idle] == 0).all()
            flags.zero_()
    touched = (dense["m"] != 0).any(dim=1)
    assert touched.any() and (~touched).any()
    assert torch.equal(dense["p"][~touched], p0[~touched])      # a row that never had gradient has zero moments and does not move


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_engine_row_bookkeeping_equals_dense_over_many_steps(dtype):
    """the training step with the row-wise bookkeeping against the same step with the dense zero-fill / norm / Adam: 10 updates on
    changing batches, an evaluation pass and a second forward + backward without an update in between; parameters and moments agree
    to the run-to-run noise of the step itself (float atomics in the gradient products)"""
    from variational_mmt_amd.engine import Dims, Engine
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True)
    p = O.init_params(c, seed=2)
    engines = {}
    for rows in (True, False):
        e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype=dtype, device="cuda", seed=1)
        e.row_adam = rows                       # (opt-in: set before the first forward builds the plans)
        assert e.rows_active() == rows
        e.load_state_dict(p)
        engines[rows] = e
    for step in range(10):
        bt = O.synth_batch(c, 6, 5 + step % 3, 6 + step % 2, n_img=12, seed=50 + step, fixed_len=False)
        for e in engines.values():
            e.set_image_table(bt["table"])
            if step == 4:        # a forward + backward whose gradients are thrown away (no update), then the real one
                other = O.synth_batch(c, 6, 7, 6, n_img=12, seed=999, fixed_len=False)
                ws = e.forward(other["src"], other["src_len"], other["tgt"], other["indices"], training=True, eps=other["eps"])
                e.loss_backward(ws, normalization=6)
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
            e.loss_backward(ws, normalization=6)
            e.optim_step(lr=0.01 if step < 6 else 0.004, max_grad_norm=5.0 if step % 4 else 0.5)
            if step == 5:        # an evaluation pass between updates (validation inside an epoch)
                e.loss(e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=False))
    torch.cuda.synchronize()
    a, b = engines[True], engines[False]
    tol = 5e-6 if dtype == "f32" else 2e-3
    for x, y, what in ((a.flat_p, b.flat_p, "p"), (a.flat_m, b.flat_m, "m"), (a.flat_v, b.flat_v, "v")):
        n = a.n_opt
        err = (x[:n] - y[:n]).abs().max().item()
        assert err <= tol * max(1.0, y[:n].abs().max().item()), (what, err)
    assert all(int(t["flags"].sum()) == 0 for t in a.row_tables)
    names = [en[2] for en in a.workspace(6, 5, 5).plan_fwd_train]
    assert names.count("vmmt_rows_mark") == 2 and names.count("vmmt_rows_zero") == 2
    assert "vmmt_rows_mark" not in [en[2] for en in a.workspace(6, 5, 5).plan_fwd_eval]


def test_row_bookkeeping_switched_on_and_off_between_updates():
    """`Engine.row_adam` may be set after construction: the cached launch plans carry or omit the row entries, and with the
    bookkeeping off the dense kernels expect fully cleared table gradients -- a change drops the plans and clears the tables' gradients
    and flags.  Six updates with the switch flipped twice against six dense updates."""
    from variational_mmt_amd.engine import Dims, Engine
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True)
    p = O.init_params(c, seed=2)
    a = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", seed=1)
    b = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", seed=1)
    for e in (a, b):
        e.load_state_dict(p)
    for step in range(6):
        if step == 2:
            a.row_adam = True
            assert a.rows_active() and not a.ws                      # plans dropped
        if step == 4:
            a.row_adam = False
            assert not a.rows_active() and all(int(t["flags"].sum()) == 0 for t in a.row_tables)
        bt = O.synth_batch(c, 6, 5 + step % 3, 6 + step % 2, n_img=12, seed=70 + step, fixed_len=False)
        for e in (a, b):
            e.set_image_table(bt["table"])
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
            e.loss_backward(ws, normalization=6)
            e.optim_step(lr=0.01, max_grad_norm=0.5)
    torch.cuda.synchronize()
    n = a.n_opt
    assert (a.flat_p[:n] - b.flat_p[:n]).abs().max().item() <= 5e-6 * max(1.0, b.flat_p[:n].abs().max().item())
    assert (a.flat_m[:n] - b.flat_m[:n]).abs().max().item() <= 5e-6 * max(1.0, b.flat_m[:n].abs().max().item())
