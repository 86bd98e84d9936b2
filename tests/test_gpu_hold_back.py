"""The side-stream half of an optimiser step held back until the next forward's head is through (Engine.bg_after_head + hold_back, which the
owner of a training loop sets: TrainerMultimodal._train_loop, bench.py).  What must hold:

  * same trajectory as the engine that issues every update at once (onmt/Optim.py:78-96: one clipped Adam step per batch);
  * `engine.params[...]`, `flat_p / flat_m / flat_v`, `state_dict()` after optim_step() are the UPDATED values (reading them issues the
    held-back half first and orders the current stream behind it);
  * a forward in eval mode, a second update without a forward, the guard of a timed-out recurrence: all see a consistent model;
  * the plan layout that goes with it (source rows + the encoder's first input projection in front, generator's third of the update and
    the gradient zeroing on the AUX stream) is also correct for one-layer models and without holding back."""
import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu
LR = 0.002


def _engine(c, p, after_head, hold, dtype="bf16", rows=False):
    from variational_mmt_amd.engine import Dims, Engine
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype=dtype, device="cuda", seed=3)
    e.bg_after_head = after_head
    e.hold_back = hold
    e.row_adam = rows                                 # (the lazy update of the embedding tables: the held-back half carries the target table's share)
    assert e.rows_active() == rows
    e.load_state_dict(p)
    return e


def _step(e, bt, B):
    ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
    e.loss_backward(ws, normalization=B)
    e.optim_step(lr=LR, max_grad_norm=5.0)
    return ws


def _close(a, b, n_updates):
    """two runs of the same updates differ by their float atomics; Adam turns an element whose gradient is rounding noise into +- lr"""
    d = (a - b).abs()
    assert d.max().item() <= n_updates * LR * 1.01 and d.mean().item() <= 2e-6 * n_updates, (d.max().item(), d.mean().item())


@pytest.mark.parametrize("layers,brnn,dtype,rows", [(2, False, "bf16", False), (1, True, "bf16", False), (2, False, "f32", False),
                                                    (2, False, "bf16", True)])
def test_held_back_update_is_the_same_update(layers, brnn, dtype, rows):
    c = O.Cfg(vs=61, vt=300, emb=64, hid=256, z=128, layers=layers, brnn=brnn)
    p = O.init_params(c, seed=1)
    B = 32
    bts = [O.synth_batch(c, B=B, S=7, T=8, n_img=40, seed=60 + i, fixed_len=False) for i in range(4)]
    ref = _engine(c, p, False, False, dtype)         # every update issued at once, the classic plan layout, dense Adam
    lay = _engine(c, p, True, False, dtype, rows)    # the new plan layout, nothing held back
    held = _engine(c, p, True, True, dtype, rows)    # held back
    for e in (ref, lay, held):
        e.set_image_table(bts[0]["table"])
    assert "BG_FLUSH" in [en[2] for en in _step(held, bts[0], B).plan_fwd_train]
    assert held._pending_bg and len(held._pending_bg) == 2
    _step(ref, bts[0], B)
    ws = _step(lay, bts[0], B)
    assert "BG_FLUSH" in [en[2] for en in ws.plan_fwd_train] and lay._pending_bg is None
    # reading a parameter of the decoder side issues the held-back half
    name = "decoder.rnn.weight_hh_l0"
    got = held.params[name].clone()
    assert held._pending_bg is None
    torch.cuda.synchronize()
    _close(got, ref.params[name], 1)
    assert (got - p[name].to(got.device)).abs().max().item() > 0.5 * LR          # ... and it HAS been updated
    for i in (1, 2, 3):
        for e in (ref, lay, held):
            _step(e, bts[i], B)
    assert held._pending_bg
    # eval-mode forward with a held-back half: its plan carries no marker, everything goes out first
    ws = held.forward(bts[0]["src"], bts[0]["src_len"], bts[0]["tgt"], bts[0]["indices"], training=False)
    assert held._pending_bg is None
    wr = ref.forward(bts[0]["src"], bts[0]["src_len"], bts[0]["tgt"], bts[0]["indices"], training=False)
    torch.cuda.synchronize()
    assert (ws.AH.view().float() - wr.AH.view().float()).abs().max().item() <= (2e-2 if dtype == "bf16" else 1e-3)
    n = ref.n_opt
    for e in (lay, held):
        _close(e.flat_p[:n], ref.flat_p[:n], 4)
        _close(e.flat_m[:n], ref.flat_m[:n], 4)
        assert e.step_count == ref.step_count == 4
    sd = held.state_dict()
    _close(sd["generator.0.weight"], ref.state_dict()["generator.0.weight"], 4)


def test_two_updates_without_a_forward_and_the_guard():
    c = O.Cfg(vs=61, vt=300, emb=64, hid=256, z=128, layers=2, brnn=False)
    p = O.init_params(c, seed=1)
    B = 32
    bt = O.synth_batch(c, B=B, S=7, T=8, n_img=40, seed=61, fixed_len=False)
    e = _engine(c, p, True, True)
    e.set_image_table(bt["table"])
    _step(e, bt, B)
    p1 = e.flat_p.clone()                  # (issues the held-back half)
    # the report of a timed-out recurrence arrives before the update: BOTH halves must skip, also the one that is issued later
    ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
    e.loss_backward(ws, normalization=B)
    e._guard[0] = 0x300
    e.optim_step(lr=LR, max_grad_norm=5.0)
    assert e._pending_bg
    e.seq_fallback = True
    e.check_async_errors()                 # issues the held-back half (guard still set), then settles the guard
    assert e._pending_bg is None and torch.equal(e.flat_p, p1) and e.step_count == 1 and e.steps_skipped == 1
    # two updates in a row on the same gradients (no forward between them): the first one's held-back half goes out before the second
    twin = _engine(c, p, False, False)
    twin.set_image_table(bt["table"])
    twin.persistent_lstm = e.persistent_lstm
    twin.load_state_dict(e.state_dict())
    twin.flat_m.copy_(e.flat_m); twin.flat_v.copy_(e.flat_v); twin.step_count = e.step_count
    for eng in (e, twin):
        ws = eng.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
        eng.loss_backward(ws, normalization=B)
        eng.optim_step(lr=LR, max_grad_norm=5.0)
        eng._sumsq_by_plan = True          # (the norm of the same gradients: as accumulated by the backward plan)
        eng.optim_step(lr=LR, max_grad_norm=5.0)
    n = e.n_opt
    _close(e.flat_p[:n], twin.flat_p[:n], 2)
    assert e.step_count == twin.step_count == 3


def test_trainer_loop_sets_and_clears_hold_back():
    import inspect
    from variational_mmt_amd.onmt import TrainerMultimodal as T
    src = inspect.getsource(getattr(T, "TrainerMultimodal", T)._train_loop)
    assert "hold_back = True" in src and "hold_back = False" in src and "wait_background()" in src
