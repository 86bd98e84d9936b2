"""A TRAJECTORY, not a step: 50 consecutive updates -- forward, ELBO, backward, clip + Adam
(onmt/TrainerMultimodal.py:679-711, onmt/Optim.py:78-96) -- at hid 128 / V 2000 / batch 32 on five recurring batches, dropout off,
the sample eps injected:

  * fp32 parity mode against the CPU oracle's own loop (`O.step_grads` + `O.clip_and_adam`): the ELBO of every step and the parameters
    at steps 1 / 5 / 10 / 25 / 50.  Two correct implementations of this loop drift apart: Adam divides by sqrt(v), so an element whose
    gradient is small against the rounding noise of its sum can move by up to 2 lr per step in either direction, and the difference
    feeds back through the next forward.  The bound therefore GROWS with the step count k (stated below: MAX_DP(k)); what it must
    catch is a systematic error -- a wrong bias correction, a clip that is applied twice, moments that are not carried -- which
    shows up as a difference of the order of lr * k, two orders of magnitude above it;
  * bf16 throughput mode against fp32 mode: the ELBO curve within 1 % at every step (measured: 1.3e-4);
  * the exact lazy Adam of the embedding tables (Engine.row_adam, the default) on and off: same trajectory."""
import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu

STEPS, LR, CLIP = 50, 0.002, 5.0
CHECK_AT = (1, 5, 10, 25, 50)


def MAX_DP(k):
    """bound on max |p_gpu - p_oracle| after k updates (fp32 mode, lr = 0.002).  Measured on MI355X: 5.8e-5 from the FIRST update on
    (one element of decoder.rnn.weight_ih_l0 whose gradient is of the order of Adam's eps = 1e-9: m / (sqrt(v) + eps) turns rounding
    noise into 0.03 lr; 9.5e-5 on another box), 6.0e-5 / 9.8e-5 after 50.  Bound: 2e-4 + 2e-6 k, i.e. 0.1 -> 0.15 lr; a systematic error
    is ~ lr k = 0.1"""
    return 2e-4 + 2e-6 * k


def MEAN_DP(k):
    """bound on mean |p_gpu - p_oracle| over the optimised parameters: measured 2.7e-10 (k = 1) ... 6.2e-9 (k = 50; 1.5e-8 in the run in which
    the ill-conditioned class walked), i.e. 3e-6 lr after fifty updates; bound 2e-9 + 1e-9 k"""
    return 2e-9 + 1e-9 * k


def MAX_DELBO(k):
    """relative bound on |ELBO_gpu - ELBO_oracle| at update k (fp32 mode): measured <= 2.5e-7 over the 50 updates; bound 1e-5 + 1e-6 k
    (the single-step tolerance of the parity tests is 2e-5)"""
    return 1e-5 + 1e-6 * k


def _engine(c, p, dtype, rows=False):
    from variational_mmt_amd.engine import Dims, Engine
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype=dtype, device="cuda", seed=1)
    e.row_adam = rows
    assert e.rows_active() == rows
    e.load_state_dict(p)
    return e


def test_fifty_updates_against_the_oracle_loop():
    c = O.Cfg(vs=2000, vt=2000, emb=64, hid=128, z=32, layers=1, brnn=True)
    p0 = O.init_params(c, seed=4)
    B = 32
    bts = [O.synth_batch(c, B, 9 + i % 3, 10 + i % 2, n_img=64, seed=300 + i, fixed_len=False) for i in range(5)]
    table = bts[0]["table"]
    engines = {"f32": _engine(c, p0, "f32"), "f32_rows": _engine(c, p0, "f32", rows=True),
               "bf16": _engine(c, p0, "bf16"), "bf16_rows": _engine(c, p0, "bf16", rows=True)}
    for e in engines.values():
        e.set_image_table(table)
    # ---- the oracle's loop on the CPU
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    p, state = {k: v.clone() for k, v in p0.items()}, {}
    elbo_o, snap_o = [], {}
    for k in range(1, STEPS + 1):
        bt = bts[(k - 1) % len(bts)]
        _, Lo, g = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], table[bt["indices"]], bt["eps"])
        elbo_o.append(float(Lo["elbo"]))
        p, _ = O.clip_and_adam(p, g, state, lr=LR, max_grad_norm=CLIP)
        if k in CHECK_AT:
            snap_o[k] = {n: v.clone() for n, v in p.items()}
    assert elbo_o[-1] < 0.9 * elbo_o[0]                  # the loop really trains (five batches, fifty updates)
    # ---- the same loop on the device
    elbo, snaps = {n: [] for n in engines}, {n: {} for n in engines}
    for k in range(1, STEPS + 1):
        bt = bts[(k - 1) % len(bts)]
        for n, e in engines.items():
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
            e.loss_backward(ws, normalization=B)
            elbo[n].append(ws.stats.clone())
            e.optim_step(lr=LR, max_grad_norm=CLIP)
            if k in CHECK_AT and n.startswith("f32"):
                snaps[n][k] = e.state_dict()
    torch.cuda.synchronize()
    import variational_mmt_amd._lib as L

    def curve(n):
        out = []
        for s in torch.stack(elbo[n]).tolist():
            out.append(s[L.STAT_NLL] - s[L.STAT_IMG_LOGPROB] + s[L.STAT_KL_SUM] / B)
        return out
    cur = {n: curve(n) for n in engines}
    report = []
    # fp32 mode against the oracle: ELBO at every step, parameters at the check points
    for k in range(1, STEPS + 1):
        rel = abs(cur["f32"][k - 1] - elbo_o[k - 1]) / abs(elbo_o[k - 1])
        assert rel <= MAX_DELBO(k), ("ELBO fp32 vs oracle", k, rel, MAX_DELBO(k))
    # the ill-conditioned class of DESIGN.md section 2 (image-network fc1 / gate: after H1 the image term does not depend on the scale of
    # mu_v, so these gradients are the residue of a 2048-term cancellation, ~1e-6 of the text path's): Adam divides such a gradient by
    # its own magnitude, so where it is rounding noise an element can walk a fraction of lr per update in either direction -- in two
    # runs of the SAME engine as much as against the oracle (seen: 3.9e-4 = 0.2 lr after 50 updates in one run of three).  Bound: 1 lr.
    ILL = ("inf_net_image.location.fc1", "inf_net_image.gate_affine_transform")
    ill_worst = 0.0
    for k in CHECK_AT:
        worst, where = 0.0, None
        for n, ref in snap_o[k].items():
            if n not in engines["f32"].grads:
                continue              # (never optimised: inf_net_image.scale.*, H6)
            d = (snaps["f32"][k][n].cpu() - ref).abs().max().item()
            if n.startswith(ILL):
                ill_worst = max(ill_worst, d)
                continue
            if d > worst:
                worst, where = d, n
        tot = sum(float((snaps["f32"][k][n].cpu() - ref).abs().sum()) for n, ref in snap_o[k].items() if n in engines["f32"].grads)
        cnt = sum(ref.numel() for n, ref in snap_o[k].items() if n in engines["f32"].grads)
        report.append((k, worst, where, tot / cnt))
    print("trajectory: max / mean |dp| vs oracle at steps", [(k, "%.2e" % w, wh, "%.2e" % mean) for k, w, wh, mean in report],
          "| max rel dELBO fp32", "%.2e" % max(abs(a - b) / abs(b) for a, b in zip(cur["f32"], elbo_o)),
          "| bf16 vs fp32", "%.2e" % max(abs(a - b) / abs(b) for a, b in zip(cur["bf16"], cur["f32"])))
    for k, worst, where, mean in report:
        assert worst <= MAX_DP(k) and mean <= MEAN_DP(k), ("parameters fp32 vs oracle", k, worst, where, mean, MAX_DP(k), MEAN_DP(k))
    assert ill_worst <= LR, ("ill-conditioned image-network class", ill_worst)
    # bf16 against fp32: within 1 % at every step
    for k in range(STEPS):
        assert abs(cur["bf16"][k] - cur["f32"][k]) <= 1e-2 * abs(cur["f32"][k]), ("ELBO bf16 vs fp32", k + 1, cur["bf16"][k], cur["f32"][k])
        assert abs(cur["bf16_rows"][k] - cur["bf16"][k]) <= 2e-3 * abs(cur["bf16"][k]), ("ELBO bf16 rows vs dense", k + 1)
        assert abs(cur["f32_rows"][k] - cur["f32"][k]) <= MAX_DELBO(k + 1) * abs(cur["f32"][k]), ("ELBO f32 rows vs dense", k + 1)
    # lazy rows on / off: the same trajectory (the update is bit-identical; two runs differ by their float atomics only)
    a, b = engines["f32_rows"], engines["f32"]
    d_rows = (a.flat_p[:a.n_opt] - b.flat_p[:b.n_opt]).abs()
    for n in a.grads:                                        # (the same split: the ill-conditioned class against 1 lr)
        if n.startswith(ILL):
            o_, shp = a.offsets[n]
            cnt = 1
            for x in shp:
                cnt *= x
            assert d_rows[o_:o_ + cnt].max().item() <= LR, n
            d_rows[o_:o_ + cnt] = 0
    assert d_rows.max().item() <= MAX_DP(STEPS)
    assert engines["f32"].step_count == STEPS and a.lazy_errors() == [0, 0]
