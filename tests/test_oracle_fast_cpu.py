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
