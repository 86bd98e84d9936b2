"""The CPU-baseline variant of the oracle (torch's fused LSTM, oracle/fast_cpu.py) equals the plain restatement."""
import pytest
import torch

from oracle import fast_cpu as F
from oracle import vi1_oracle as O


@pytest.mark.parametrize("brnn,layers,fixed", [(True, 1, False), (False, 2, False), (True, 2, True)])
def test_fused_lstm_equals_restatement(brnn, layers, fixed):
    c = O.Cfg(vs=53, vt=47, emb=12, hid=16, z=8, img=2048, layers=layers, brnn=brnn)
    p = O.init_params(c, seed=4, dtype=torch.float64)
    bt = O.synth_batch(c, 7, 6, 8, n_img=9, seed=2, fixed_len=fixed, dtype=torch.float64)
    img = bt["table"][bt["indices"]]
    r0, L0, g0 = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"])
    r1, L1, g1 = F.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"])
    for k in ("context", "rnn_out", "attn", "mu", "sigma", "mu_v", "enc_h_n", "enc_c_n"):
        assert (r0[k] - r1[k]).abs().max().item() < 1e-9, k
    assert abs(float(L0["elbo"]) - float(L1["elbo"])) < 1e-8
    assert set(g0) == set(g1)
    for k in g0:
        assert (g0[k] - g1[k]).abs().max().item() <= 1e-6 * max(1e-12, g0[k].abs().max().item()), k


@pytest.mark.parametrize("scale", [0.5, 400.0])       # without / with the norm clip engaged
def test_in_place_adam_equals_restatement(scale):
    """the optimiser leg of the timed CPU baseline (in-place, torch.optim.Adam's operation order) against vi1_oracle.clip_and_adam,
    three consecutive steps"""
    c = O.Cfg(vs=53, vt=47, emb=12, hid=16, z=8, img=2048, layers=1, brnn=True)
    p0 = O.init_params(c, seed=4)
    g = torch.Generator().manual_seed(5)
    pa = {k: v.clone() for k, v in p0.items()}
    pb = {k: v.clone() for k, v in p0.items()}
    sa, sb = {}, {}
    for _ in range(3):
        grads = {k: (torch.rand(v.shape, generator=g) - 0.5) * 1e-2 * scale for k, v in p0.items() if "inf_net_image.scale" not in k}
        pa, na = O.clip_and_adam(pa, {k: v.clone() for k, v in grads.items()}, sa)
        pb, nb = F.clip_and_adam(pb, {k: v.clone() for k, v in grads.items()}, sb)
        assert abs(na - nb) <= 2e-4 * na and (na > 5.0) == (scale > 1)       # (fp32 per-tensor norms, like clip_grad_norm, against the fp64 sum)
    for k in pa:
        assert (pa[k] - pb[k]).abs().max().item() <= 5e-6, k
