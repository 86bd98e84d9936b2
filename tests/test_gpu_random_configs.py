"""Randomised configurations (seeded: the same 40 every run): the fp32 parity mode of the HIP step against the CPU oracle's forward,
loss and autograd on shapes nobody picked by hand -- one sentence, one source position, two target positions, hidden sizes that are
not multiples of anything the kernels tile, one or two layers, uni- / bidirectional, ragged lengths, the conditional prior, free bits,
KL annealing, dropout with the device's masks injected, token normalisation.  Tolerances of tests/test_gpu_step_parity.py (fp32:
statistics 3e-5 relative, gradients 3e-4 of the tensor's max; the ill-conditioned image-network class 5e-2)."""
import os
import random

import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu
ILL = ("inf_net_image.location.fc1", "inf_net_image.gate_affine_transform")


def _draw(seed):
    r = random.Random(1000 + seed)
    brnn = r.random() < 0.5
    conditional = r.random() < 0.25
    hid = r.choice([6, 10, 12, 20, 24, 34, 48, 64])
    if brnn or conditional:
        hid += hid % 2                      # the directions split the hidden size (the conditional model's encoder_tgt is bidirectional)
    layers = r.choice([1, 1, 2])
    c = O.Cfg(vs=r.randint(8, 60), vt=r.randint(9, 70), emb=r.choice([5, 8, 12, 17, 32]), hid=hid, z=r.choice([3, 4, 8, 13, 16]),
              layers=layers, brnn=brnn, conditional=conditional)
    B = r.choice([1, 2, 3, 5, 8, 13, 33])
    S = r.choice([1, 2, 3, 7, 11])
    T = r.choice([2, 3, 4, 8, 12])
    opts = dict(dropout=r.random() < 0.35, freebits=r.random() < 0.3, kl_mult=r.choice([1.0, 1.0, 0.37]), tokens=r.random() < 0.3,
                fixed_len=r.random() < 0.3 or T < 3)          # (the oracle's ragged generator draws target lengths from 3 up)
    return c, B, S, T, opts


@pytest.mark.parametrize("seed", range(int(os.environ.get("VMMT_TEST_SEEDS_F32", "40"))))        # (bug hunts: VMMT_TEST_SEEDS_F32=400)
def test_random_configuration_fp32_against_the_oracle(seed):
    from variational_mmt_amd.engine import Dims, Engine
    c, B, S, T, o = _draw(seed)
    p = O.init_params(c, seed=seed)
    bt = O.synth_batch(c, B, S, T, n_img=max(B, 4) + 3, seed=500 + seed, fixed_len=o["fixed_len"])
    drop = 0.5 if o["dropout"] else 0.0
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, drop, conditional=c.conditional), dtype="f32", device="cuda", seed=seed)
    e.load_state_dict(p)
    e.set_image_table(bt["table"])
    n_tok = int((bt["tgt"][1:] != 1).sum())
    norm = float(n_tok if o["tokens"] else B)
    margin = 0.0
    img = bt["table"][bt["indices"]]
    tl = bt["tgt_len"] if c.conditional else None
    if o["freebits"]:                                   # a margin on one side or the other of the batch-mean KL
        with torch.no_grad():
            r0 = O.forward(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], True, None, False, tgt_len=tl)
            kl0 = float(O.loss(p, c, r0, bt["tgt"], img)["kl_before"]) * o["kl_mult"]
        margin = kl0 * (2.0 if seed % 2 else 0.5)
    ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"], tgt_len=tl)
    e.loss_backward(ws, normalization=norm, kl_mult=o["kl_mult"], use_freebits=o["freebits"], margin=margin)
    torch.cuda.synchronize()
    masks = None
    if drop > 0:
        H, Tp = c.hid, T - 1
        masks = {"dec_out": ws.out_mask.view().float().cpu().view(Tp, B, H)}
        for l in range(c.layers - 1):
            masks["enc_l%d" % l] = ws.enc_mask[l].view().float().cpu().view(S, B, H)
            masks["dec_l%d" % l] = ws.dec_mask[l].view().float().cpu().view(Tp, B, H)
            if c.conditional:
                # encoder_tgt walks the BATCH axis (H5): its activations are [B (time)][T (batch)][fwd : htp | bwd : htp], rows b T + t
                mk = ws.enct_mask[l].view().float().cpu()
                ht, htp = e.d.ht, e.d.htp
                masks["enct_l%d" % l] = torch.cat([mk[:, :ht], mk[:, htp:htp + ht]], 1).reshape(B, T, 2 * ht)
    r, Lo, g = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], masks=masks, normalization=norm, kl_mult=o["kl_mult"],
                            use_freebits=o["freebits"], freebits=margin, tgt_len=tl)
    st = e.read_stats(ws, kl_mult=o["kl_mult"], use_freebits=o["freebits"], margin=margin)
    what = (seed, vars(c) if hasattr(c, "__dict__") else c, B, S, T, o)
    for k, ok in (("nmt", "nll"), ("td_kl_before", "kl_before"), ("td_kl_after", "kl_after"), ("elbo", "elbo"), ("img_feats_loss", "img_logprob")):
        ref = float(Lo[ok])
        assert abs(st[k] - ref) <= 3e-5 * max(abs(ref), 1e-3), (k, st[k], ref, what)
    assert st["n_words"] == Lo["n_words"] == n_tok and abs(st["n_correct"] - Lo["n_correct"]) <= 1, what
    assert set(g) == set(e.grads), what
    for k in g:
        got, ref = e.grads[k].cpu().double(), g[k].double()
        scale = max(ref.abs().max().item(), 1e-6)
        tol = 5e-2 if k.startswith(ILL) else 3e-4          # (the cancellation class: 1.5 % seen on a gradient of 1e-6 in 400 draws)
        assert (got - ref).abs().max().item() <= tol * scale + 1e-9, (k, (got - ref).abs().max().item(), scale, what)


def _draw_fast(seed):
    r = random.Random(7000 + seed)
    brnn = r.random() < 0.5
    hid = r.choice([64, 128, 256, 512, 500, 96])        # incl. the scripts' 500 (padded compute layout) and a size the fast kernels skip
    layers = r.choice([1, 1, 2])
    if brnn and (hid // 2) % 32:
        brnn = False
    c = O.Cfg(vs=r.randint(50, 900), vt=r.choice([300, 777, 2000, 3001]), emb=r.choice([64, 100, 256]), hid=hid, z=r.choice([32, 128, 256, 100]),
              layers=layers, brnn=brnn)
    B = r.choice([7, 32, 40, 100, 129])
    S = r.choice([3, 9, 20, 31])
    T = r.choice([4, 10, 21, 33])
    return c, B, S, T, dict(dropout=r.random() < 0.4, fixed_len=r.random() < 0.4, tokens=r.random() < 0.3)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VMMT_TEST_SEEDS_BF16", "16"))))
def test_random_configuration_bf16_against_the_oracle(seed):
    """the bf16 throughput mode on random shapes that select (or just miss) the fast kernels -- persistent recurrences, fused q(z|x), fused
    sweep with and without compaction, MFMA attention, grouped weight gradients, padded hidden sizes -- against the oracle: statistics at
    the bf16 tolerance, every gradient tensor by relative L2"""
    from variational_mmt_amd.engine import Dims, Engine
    c, B, S, T, o = _draw_fast(seed)
    p = O.init_params(c, seed=seed)
    bt = O.synth_batch(c, B, S, T, n_img=B + 5, seed=900 + seed, fixed_len=o["fixed_len"])
    drop = 0.5 if o["dropout"] else 0.0
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, drop), dtype="bf16", device="cuda", seed=seed)
    e.load_state_dict(p)
    e.set_image_table(bt["table"])
    n_tok = int((bt["tgt"][1:] != 1).sum())
    norm = float(n_tok if o["tokens"] else B)
    img = bt["table"][bt["indices"]]
    ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
    e.loss_backward(ws, normalization=norm)
    torch.cuda.synchronize()
    assert not any(e.lstm_seq_errors())
    masks = None
    if drop > 0:
        H, Tp = c.hid, T - 1
        masks = {"dec_out": ws.out_mask.view().float().cpu().view(Tp, B, H)}
        for l in range(c.layers - 1):
            masks["enc_l%d" % l] = ws.enc_mask[l].view().float().cpu().view(S, B, H)
            masks["dec_l%d" % l] = ws.dec_mask[l].view().float().cpu().view(Tp, B, H)
    r, Lo, g = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"], masks=masks, normalization=norm)
    st = e.read_stats(ws)
    what = (seed, c, B, S, T, o, dict(gen_fused=ws.gen_fused, fused_q=ws.fused_q, Mc=ws.gen_Mc, M=ws.M))
    for k, ok, tol in (("nmt", "nll", 5e-3), ("td_kl_before", "kl_before", 2e-2), ("elbo", "elbo", 5e-3)):
        ref = float(Lo[ok])
        assert abs(st[k] - ref) <= tol * abs(ref), (k, st[k], ref, what)
    assert st["n_words"] == Lo["n_words"] == n_tok, what
    gmax = max(v.abs().max().item() for v in g.values())
    for k in g:
        got, ref = e.grads[k].cpu().double(), g[k].double()
        if ref.abs().max().item() < 1e-4 * gmax:
            continue                     # (a tensor whose whole gradient is four orders below the step's largest: bf16 noise of its inputs)
        rel = ((got - ref).norm() / ref.norm()).item()
        assert rel <= (0.15 if k.startswith(ILL) else 0.1), (k, rel, what)      # (7.1 % seen on the scale network's fc1 at B 7, S 3 in 160 draws)
