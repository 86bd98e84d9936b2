"""GPU: the exact lazy Adam of the embedding tables (csrc/optim.hip: vmmt_rows_mark / vmmt_rows_catchup / vmmt_sumsq_rows /
vmmt_adam_rows_step; engine._build_row_tables) against the dense path.

Reference: loss.backward() into zero-filled .grad tensors, clip_grad_norm and torch.optim.Adam over EVERY element at every step
(onmt/TrainerMultimodal.py:628-629, onmt/Optim.py:68-70,94-96).  A row without gradient still moves under dense Adam (its moments
decay, the parameter follows them); the lazy path applies exactly those zero-gradient steps later -- when a batch next looks the row up,
when the rolling 1 / roll of the table comes round, or at a flush -- so parameters AND moments must come out BIT-identical.  Checked at
the kernel level, where both paths can be fed identical gradients (the step's own gradients carry float-atomic noise from run to run)."""
import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,Cc,roll,steps,max_ids", [(301, 500, 7, 60, 40), (157, 1024, 16, 60, 40), (64, 24, 0, 60, 40), (911, 256, 3, 60, 40),
                                                     # the benchmark's table and batch: every 64-row block has flagged AND rolling rows, its 16 waves
                                                     # start at different times (round 6: a wave's share of the rows depended on `last`, which the
                                                     # faster waves were already writing -- a row in a few thousand lost its update)
                                                     (30011, 500, 16, 12, 5120)])
def test_lazy_rows_kernels_bit_identical_to_dense_adam(R, Cc, roll, steps, max_ids):
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3 + R)
    p0 = (torch.rand(R, Cc, generator=g) - 0.5).to(dev)
    dense = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0))
    lazy = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0))
    grad_d, grad_l = torch.zeros_like(p0), torch.full_like(p0, 77.0)      # (the lazy path never reads a gradient row it has not cleared itself)
    flags = torch.zeros(2 * R, dtype=torch.int32, device=dev)        # [parity of the update][row]
    last = torch.zeros(R, dtype=torch.int32, device=dev)
    hist = torch.zeros(L.LAZY_HIST_WORDS, dtype=torch.int32, device=dev)
    rowsq = torch.zeros(R, dtype=torch.float32, device=dev)
    sq_d, sq_l = (torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=dev) for _ in range(2))
    skip = torch.zeros(2, dtype=torch.int32, device=dev)
    b1, b2, eps = 0.9, 0.999, 1e-9
    lz = [lazy[k].data_ptr() for k in ("p", "m", "v")]

    def catchup(mode):
        L.check(lib.vmmt_rows_catchup(lz[0], grad_l.data_ptr() if mode == 0 else None, lz[1], lz[2], R, Cc, flags.data_ptr() if mode == 0 else None,
                                      last.data_ptr(), hist.data_ptr(), b1, b2, eps, mode, st), "catchup")

    def same(tag):
        torch.cuda.synchronize()
        for k in ("p", "m", "v"):
            assert torch.equal(dense[k], lazy[k]), (tag, k, (dense[k] - lazy[k]).abs().max().item())

    n_skipped = 0
    for step in range(1, steps + 1):
        lr = 0.002 if step < 25 else 0.001                      # a learning-rate decay in the middle (replays read the step's scalars from the ring)
        max_norm = 5.0 if step % 3 else 0.02                    # clipping active on every third step
        hot = step % 11 < 8                                     # mostly a small hot set of rows (Zipf-like), sometimes any row
        ids = torch.randint(0, R // 8 if hot else R, (int(torch.randint(max(1, max_ids // 2), max_ids, (1,), generator=g)),), generator=g)
        if step % 7 == 0:
            ids = torch.cat([ids, ids[:3]])                     # duplicates in a batch
        rows = torch.unique(ids).to(dev)
        ids_d = ids.to(dev)
        L.check(lib.vmmt_rows_mark(ids_d.data_ptr(), ids_d.numel(), flags.data_ptr(), R, hist.data_ptr(), st), "mark")
        catchup(0)
        torch.cuda.synchronize()
        assert int(hist[0]) == step - 1 and int((flags.view(2, R)[step & 1] == step).sum()) >= rows.numel()
        assert (grad_l[rows] == 0).all()
        # the rows about to be looked up are current: equal to the dense parameters right now
        for k in ("p", "m", "v"):
            assert torch.equal(lazy[k][rows], dense[k][rows]), (step, k)
        gr = (torch.rand(rows.numel(), Cc, generator=g) - 0.5).to(dev) * (10.0 if step % 5 == 0 else 0.1)
        grad_d.zero_()
        grad_d[rows] = gr
        grad_l[rows] = gr
        sq_d[:L.SUMSQ_SLOTS].zero_()
        sq_l[:L.SUMSQ_SLOTS].zero_()
        L.check(lib.vmmt_sumsq(grad_d.data_ptr(), R * Cc, sq_d.data_ptr(), 0, st), "sumsq")
        L.check(lib.vmmt_sumsq_rows(grad_l.data_ptr(), R, Cc, flags.data_ptr(), hist.data_ptr(), rowsq.data_ptr(), sq_l.data_ptr(), 3, st), "sumsq_rows")
        torch.cuda.synchronize()
        a, b = float(sq_d[0]), float(sq_l[3])
        assert abs(a - b) <= 2e-6 * a, (step, a, b)             # the same norm up to the order of summation
        # a step the guard word skips (a recurrence that timed out): nothing moves in either path, the ring records it as skipped
        skipping = step in (13, 14, 40) or (steps < 13 and step == 7)
        skip[0] = 1 if skipping else 0
        n_skipped += skipping
        # the same clip coefficient for both (the bit-level claim is about the update, not about the norm's summation order)
        L.check(lib.vmmt_adam_step(dense["p"].data_ptr(), grad_d.data_ptr(), dense["m"].data_ptr(), dense["v"].data_ptr(), R * Cc, lr, b1, b2,
                                   eps, step, max_norm, sq_d.data_ptr(), 1.0, 0, None, skip.data_ptr(), st), "adam")
        L.check(lib.vmmt_adam_rows_step(lz[0], grad_l.data_ptr(), lz[1], lz[2], R, Cc, flags.data_ptr(), last.data_ptr(), hist.data_ptr(), lr, b1,
                                        b2, eps, step, roll, max_norm, sq_d.data_ptr(), 1.0, skip.data_ptr(), st), "rows_step")
        torch.cuda.synchronize()
        assert int(hist[0]) == step and int(skip[1]) == 2 * n_skipped
        if not skipping:
            lr_ = last
            assert (lr_[rows] == step).all()
            if roll:
                assert (lr_[torch.arange(R, device=dev) % roll == step % roll] == step).all()
                assert int((step - lr_).max()) <= 2 * roll + 1      # no row is ever further behind than the rolling period (twice: a row whose turn fell on a skipped step)
        if step in (1, 5, 17, 41, steps):                       # flush everything and compare the whole table
            catchup(1)
            same(step)
            assert (last == step).all()
    assert int(hist[1]) == 0                                    # no replay met an overwritten ring entry
    touched = (dense["m"] != 0).any(dim=1)
    assert touched.any()
    if (~touched).any():
        assert torch.equal(dense["p"][~touched], p0[~touched])  # a row that never had gradient has zero moments and does not move
    # a row untouched after an early update still moved (the decaying moments): the deferred steps are not no-ops
    assert (dense["p"] != p0).any(dim=1).float().mean().item() > 0.1


def test_lazy_rows_ring_overrun_is_reported():
    """without the rolling update a row may fall further behind than the ring of step scalars is long: the replay must say so (error
    word) instead of applying another step's scalars"""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    R, Cc = 8, 8
    p = torch.ones(R, Cc, device=dev)
    m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    flags = torch.zeros(2 * R, dtype=torch.int32, device=dev)
    last = torch.zeros(R, dtype=torch.int32, device=dev)
    hist = torch.zeros(L.LAZY_HIST_WORDS, dtype=torch.int32, device=dev)
    sq = torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=dev)
    one = torch.tensor([0], device=dev)
    for step in range(1, L.LAZY_HIST + 10):
        L.check(lib.vmmt_rows_mark(one.data_ptr(), 1, flags.data_ptr(), R, hist.data_ptr(), st), "mark")
        L.check(lib.vmmt_rows_catchup(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), R, Cc, flags.data_ptr(), last.data_ptr(), hist.data_ptr(),
                                      0.9, 0.999, 1e-9, 0, st), "catchup")
        g[0] = 1.0
        L.check(lib.vmmt_adam_rows_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), R, Cc, flags.data_ptr(), last.data_ptr(), hist.data_ptr(),
                                        0.01, 0.9, 0.999, 1e-9, step, 0, 0.0, sq.data_ptr(), 1.0, None, st), "rows_step")
    torch.cuda.synchronize()
    assert int(hist[1]) == 0
    L.check(lib.vmmt_rows_catchup(p.data_ptr(), None, m.data_ptr(), v.data_ptr(), R, Cc, None, last.data_ptr(), hist.data_ptr(), 0.9, 0.999, 1e-9, 1, st), "flush")
    torch.cuda.synchronize()
    assert int(hist[1]) != 0                                    # rows 1.. are LAZY_HIST + 9 steps behind: the first of them are gone from the ring


def _pair(c, p, dtype, **kw):
    from variational_mmt_amd.engine import Dims, Engine
    engines = {}
    for rows in (True, False):
        e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0, conditional=c.conditional), dtype=dtype, device="cuda", seed=1)
        e.row_adam = rows                       # (set before the first forward builds the plans)
        for k, v in kw.items():
            setattr(e, k, v)
        assert e.rows_active() == rows
        e.load_state_dict(p)
        engines[rows] = e
    return engines


@pytest.mark.parametrize("conditional", [False, True])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_engine_lazy_rows_equal_dense_over_many_steps(dtype, conditional):
    """the training step with the lazy tables against the same step with the dense zero-fill / norm / Adam: 24 updates on changing batches
    (rolling period 5: rows come round several times), an evaluation pass, a state_dict() in the middle, a learning-rate change and a
    second forward + backward without an update in between; parameters and moments agree to the run-to-run noise of the step itself
    (float atomics in the gradient products), and rows no batch ever used did not move at all.  `conditional`: encoder_tgt looks the shared
    target table up too -- by every target position and the pad fill of its transposed image, flagged by a launch of their own"""
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True, conditional=conditional)
    p = O.init_params(c, seed=2)
    engines = _pair(c, p, dtype, lazy_roll=5)
    used_src, used_tgt = set(), set()
    tl = lambda b_: dict(tgt_len=b_["tgt_len"]) if conditional else {}
    for step in range(24):
        bt = O.synth_batch(c, 6, 5 + step % 3, 6 + step % 2, n_img=12, seed=50 + step, fixed_len=False)
        used_src |= set(bt["src"].reshape(-1).tolist())
        used_tgt |= set((bt["tgt"] if conditional else bt["tgt"][:-1]).reshape(-1).tolist())
        for e in engines.values():
            e.set_image_table(bt["table"])
            if step == 4:        # a forward + backward whose gradients are thrown away (no update), then the real one
                other = O.synth_batch(c, 6, 7, 6, n_img=12, seed=999, fixed_len=False)
                ws = e.forward(other["src"], other["src_len"], other["tgt"], other["indices"], training=True, eps=other["eps"], **tl(other))
                e.loss_backward(ws, normalization=6)
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"], **tl(bt))
            e.loss_backward(ws, normalization=6)
            e.optim_step(lr=0.01 if step < 6 else 0.004, max_grad_norm=5.0 if step % 4 else 0.5)
            if step == 5:        # an evaluation pass between updates (validation inside an epoch): flushes
                e.loss(e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=False, **tl(bt)))
            if step == 9:
                sd = e.state_dict()                      # flushes
                assert all(torch.isfinite(v).all() for v in sd.values())
        if step == 4:            # (the thrown-away batch's rows were flagged: they take a zero-gradient step like every other row)
            used_src |= set(other["src"].reshape(-1).tolist())
            used_tgt |= set((other["tgt"] if conditional else other["tgt"][:-1]).reshape(-1).tolist())
    if conditional:
        used_tgt.add(O.PAD)
    torch.cuda.synchronize()
    a, b = engines[True], engines[False]
    assert a._lazy_dirty and a.lazy_errors() == [0, 0]
    # f32: two DENSE runs of these 24 updates differ by up to 4e-6 (float atomics in the gradient products; tests/run_to_run_matrix.py), the lazy
    # engine once by 4.95e-6 from the dense one: 2e-5 leaves that room and is two orders below what a row that missed one update would show (lr)
    tol = 2e-5 if dtype == "f32" else 2e-3
    for x, y, what in ((a.flat_p, b.flat_p, "p"), (a.flat_m, b.flat_m, "m"), (a.flat_v, b.flat_v, "v")):      # (reading the arena flushes)
        n = a.n_opt
        err = (x[:n] - y[:n]).abs()
        bound = tol * max(1.0, y[:n].abs().max().item())
        for t in a.row_tables:          # the tables themselves: to the step's own run-to-run noise
            assert err[t["off"]:t["end"]].max().item() <= bound, (what, t["name"], err[t["off"]:t["end"]].max().item())
        if dtype == "f32":
            assert err.max().item() <= bound, (what, err.max().item())
        else:
            # bf16, everything else: two DENSE runs of these 24 updates already differ by 1e-4 .. 3.4e-3 in a handful of elements of the image
            # network's fc1 (tools' run-to-run matrix, LABNOTES round 6): float-atomic noise moves an activation across a bf16 rounding step
            # or a ReLU gate, and Adam with eps 1e-9 turns a gradient of that size into a whole step of lr.  A few such elements, each
            # by at most two steps of the largest lr -- anything systematic in the row update would show in the tables above
            assert int((err > bound).sum()) <= 16 and err.max().item() <= 2e-2, (what, int((err > bound).sum()), err.max().item())
    assert not a._lazy_dirty and all(int((t["last"] != a.step_count).sum()) == 0 for t in a.row_tables)
    for name, used in (("encoder.embeddings.make_embedding.emb_luts.0.weight", used_src), ("decoder.embeddings.make_embedding.emb_luts.0.weight", used_tgt)):
        idle = [r for r in range(p[name].shape[0]) if r not in used]
        assert idle
        for e in (a, b):
            assert torch.equal(e.params[name][idle].cpu(), p[name][idle]), name
    names = [en[2] for en in a.workspace(6, 5, 5).plan_fwd_train]
    assert names.count("vmmt_rows_mark") == (1 if conditional else 0)            # (encoder_tgt's ids; everything else is flagged by vmmt_prepare_batch)
    if conditional:
        assert names.index("vmmt_rows_mark") < names.index("vmmt_rows_catchup")
    catchups = [n for n in names if n.startswith("vmmt_rows_catchup")]           # (the rows are flagged by vmmt_prepare_batch)
    # bf16: the source table's catch-up also writes the rows' bf16 copy, which the encoder's input projection reads by token id
    assert sorted(catchups) == (["vmmt_rows_catchup", "vmmt_rows_catchup_shadow"] if dtype == "bf16" else ["vmmt_rows_catchup"] * 2), catchups
    assert names.index(catchups[0]) < names.index("vmmt_gather_rows") and names.index(catchups[0]) < names.index("gemm")      # in front of the lookups
    for k in ("vmmt_rows_catchup", "vmmt_rows_catchup_shadow"):
        assert k not in [en[2] for en in a.workspace(6, 5, 5).plan_fwd_eval]
        assert k not in [en[2] for en in b.workspace(6, 5, 5).plan_fwd_train]
    if dtype == "bf16" and not conditional:
        # the gradient norm's pieces sit behind the launches that complete their gradients, on those launches' streams (Engine.tail_norm_first):
        # the tables' flagged rows right behind their scatter-adds, nothing but the two joins behind the main stream's last launch
        bw = [(en[2], en[4]) for en in a.workspace(6, 5, 5).plan_bwd]
        scat = [i for i, (n, _s) in enumerate(bw) if n == "vmmt_scatter_add_rows"]
        assert len(scat) == 2 and all(bw[i + 1] == ("SUMSQ_ROWS", bw[i][1]) for i in scat), [bw[i:i + 2] for i in scat]
        assert sorted(bw[i][1] for i in scat) == [0, 1]                       # source rows on the main stream, target rows on the side stream
        last_launch = max(i for i, (n, s_) in enumerate(bw) if s_ == 0 and n not in ("EV_WAIT", "EV_RECORD"))
        assert all(n == "EV_WAIT" for n, s_ in bw[last_launch + 1:] if s_ == 0) and bw[last_launch][0] == "SUMSQ_ROWS"
    if dtype == "bf16":          # the fused lookup: no source gather in front of the first product, one gather less on the main stream
        fused = [en for en in a.workspace(6, 5, 5).plan_fwd_train if en[2] == "gemm" and en[3].a_row_ids]
        assert len(fused) == 1 and not any(en[3].a_row_ids for en in b.workspace(6, 5, 5).plan_fwd_train if en[2] == "gemm")


def test_lazy_rows_switched_on_and_off_between_updates():
    """`Engine.row_adam` may be set after construction: the cached launch plans carry or omit the row entries; switched off,
    every row is brought up to date first and the dense kernels find cleared table gradients.  Eight updates with the switch flipped
    three times against eight dense updates."""
    from variational_mmt_amd.engine import Dims, Engine
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True)
    p = O.init_params(c, seed=2)
    a = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", seed=1)
    b = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", seed=1)
    b.row_adam = False
    assert a.rows_active() and not b.rows_active()               # the lazy tables are the default
    for e in (a, b):
        e.load_state_dict(p)
    for step in range(8):
        if step == 2:
            a.row_adam = False
            assert not a.rows_active() and not a.ws and not a._lazy_dirty      # plans dropped, rows flushed
        if step == 4:
            a.row_adam = True
            assert a.rows_active() and all(int((t["last"] != a.step_count).sum()) == 0 for t in a.row_tables)
        if step == 6:
            a.row_adam = False
        bt = O.synth_batch(c, 6, 5 + step % 3, 6 + step % 2, n_img=12, seed=70 + step, fixed_len=False)
        for e in (a, b):
            e.set_image_table(bt["table"])
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
            e.loss_backward(ws, normalization=6)
            e.optim_step(lr=0.01, max_grad_norm=0.5)
    torch.cuda.synchronize()
    n = a.n_opt
    assert (a.flat_p[:n] - b.flat_p[:n]).abs().max().item() <= 2e-5 * max(1.0, b.flat_p[:n].abs().max().item())
    assert (a.flat_m[:n] - b.flat_m[:n]).abs().max().item() <= 2e-5 * max(1.0, b.flat_m[:n].abs().max().item())


def test_lazy_rows_follow_a_step_counter_set_from_outside():
    """Optim.py's mirror zeroes the moments and the counter for a NEW Adam and sets them from a loaded optimiser state; a twin engine
    in a test copies them: the tables' bookkeeping follows (rows are brought up to date under the old counter first)"""
    from variational_mmt_amd.engine import Dims, Engine
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True)
    p = O.init_params(c, seed=2)
    a, b = _pair(c, p, "f32", lazy_roll=4).values()

    def steps(e, lo, hi):
        for step in range(lo, hi):
            bt = O.synth_batch(c, 6, 5, 6, n_img=12, seed=90 + step, fixed_len=False)
            e.set_image_table(bt["table"])
            e.loss_backward(e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"]), normalization=6)
            e.optim_step(lr=0.01, max_grad_norm=5.0)
    for e in (a, b):
        steps(e, 0, 5)
        e.flat_m.zero_()
        e.flat_v.zero_()
        e.step_count = 0                          # a new Adam (Optim.py:68-70 as the reference executes it on -train_from)
        steps(e, 5, 9)
    # a twin that takes over parameters, moments and the counter mid-run
    from variational_mmt_amd.engine import Dims, Engine
    t = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", seed=1)
    t.load_state_dict(a.state_dict())
    t.flat_m.copy_(a.flat_m)
    t.flat_v.copy_(a.flat_v)
    t.step_count = a.step_count
    for e in (a, b, t):
        steps(e, 9, 14)
    torch.cuda.synchronize()
    n = a.n_opt
    for x in (a, t):
        for what in ("flat_p", "flat_m", "flat_v"):
            u, w = getattr(x, what)[:n], getattr(b, what)[:n]
            assert (u - w).abs().max().item() <= 2e-5 * max(1.0, w.abs().max().item()), what          # (14 updates of float-atomic noise)
    assert a.lazy_errors() == [0, 0] and t.lazy_errors() == [0, 0]


@pytest.mark.parametrize("rows", [False, True])
def test_a_write_to_the_arena_from_outside_goes_behind_the_side_streams_half_of_the_update(rows):
    """`engine.flat_m.zero_()` straight after optim_step() (Optim.py's mirror does that for a NEW Adam) must land behind the half of the update
    that the side stream is still running -- with the dense tables as with the lazy ones.  The side stream is held up by a long sleep in
    front of the update here, so that the order is the engine's doing and not the box's timing (the full suite caught this once as a
    one-in-many failure of the dense twin in the test above)"""
    from variational_mmt_amd.engine import Dims, Engine
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True)
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", seed=1)
    e.row_adam = rows
    e.load_state_dict(O.init_params(c, seed=2))
    assert e.use_side_stream and e.split_optim
    for step in range(2):
        bt = O.synth_batch(c, 6, 5, 6, n_img=12, seed=40 + step, fixed_len=False)
        e.set_image_table(bt["table"])
        e.loss_backward(e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"]), normalization=6)
        if step == 1:
            with torch.cuda.stream(e.side_stream):
                torch.cuda._sleep(200_000_000)          # ~0.1 s in front of the side stream's half
        e.optim_step(lr=0.01, max_grad_norm=5.0)
    e.flat_m.zero_()
    e.flat_v.zero_()
    torch.cuda.synchronize()
    assert float(e.flat_m.abs().max()) == 0.0 and float(e.flat_v.abs().max()) == 0.0
