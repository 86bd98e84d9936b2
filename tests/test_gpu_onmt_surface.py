"""GPU: the reference's Python surface (onmt.ModelConstructor / TrainerMultimodal / NMTVIModel1LossCompute / Optim) driven
the way train_mm_vi_model1.py drives it, 3 updates from a fixed state dict, against the CPU oracle; then the checkpoint
round trip (drop_checkpoint -> make_vi_model_mmt(checkpoint) + pickled Optim)."""
import argparse
import types

import numpy as np
import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu


class _Vocab(object):
    def __init__(self, n, tgt):
        sp = ["<unk>", "<blank>"] + (["<s>", "</s>"] if tgt else [])
        self.itos = sp + ["w%d" % i for i in range(n - len(sp))]
        self.stoi = {w: i for i, w in enumerate(self.itos)}

    def __len__(self):
        return len(self.itos)


class _Batch(object):
    def __init__(self, bt, dev):
        self.src = (bt["src"].to(dev), bt["src_len"].to(dev))
        self.tgt = (bt["tgt"].to(dev), bt["tgt_len"].to(dev))
        self.indices = bt["indices"].to(dev)
        self.batch_size = bt["src"].shape[1]


def _opt(c, tmp):
    return argparse.Namespace(model_type="text", multimodal_model_type="vi-model1", use_posterior_image_features=False,
                              use_global_image_features=True, path_to_train_img_feats="resnet50.hdf5", rnn_type="LSTM",
                              global_attention="general", copy_attn=False, coverage_attn=False, context_gate=None,
                              share_embeddings=False, share_decoder_embeddings=False, word_dropout=0.0, enc_layers=c.layers,
                              dec_layers=c.layers, src_word_vec_size=c.emb, tgt_word_vec_size=c.emb, brnn=c.brnn,
                              encoder_type="brnn" if c.brnn else "rnn", rnn_size=c.hid, z_latent_dim=c.z, dropout=0.0,
                              param_init=0.1, conditional=False, image_loss="logprob", compute_dtype="f32", gpuid=[0], seed=-1,
                              early_stopping_criteria="perplexity", evaluate_every_n_model_updates=500,
                              save_model=str(tmp / "ckpt"))


def test_three_updates_through_the_onmt_surface_and_checkpoint_roundtrip(tmp_path):
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    c = O.Cfg(vs=43, vt=47, emb=16, hid=32, z=8, img=2048, layers=1, brnn=True)
    p = O.init_params(c, seed=4)
    B, S, T = 7, 6, 8
    batches = [O.synth_batch(c, B, S, T, n_img=20, seed=100 + i, fixed_len=False) for i in range(3)]
    table = batches[0]["table"]
    fields = {"src": types.SimpleNamespace(vocab=_Vocab(c.vs, False)), "tgt": types.SimpleNamespace(vocab=_Vocab(c.vt, True))}
    opt = _opt(c, tmp_path)
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, None)
    assert opt.global_image_features_dim == 2048
    assert sorted(model.state_dict().keys()) == sorted(O.param_shapes(c).keys())
    model.load_state_dict({k: v for k, v in p.items()})
    train_loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab, use_kl_annealing=True,
                                                    kl_annealing_current=0.5, kl_annealing_increment=0.25, kl_annealing_warmup_steps=1)
    valid_loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    trainer = onmt.TrainerMultimodal(model, train_loss, valid_loss, optim, 0, 32, "text", "sents", 1, train_img_feats=table.numpy(),
                                     valid_img_feats=table.numpy(), multimodal_model_type="vi-model1", model_opt=opt, fields=fields)
    po, state = {k: v.clone() for k, v in p.items()}, {}
    mults = [0.5, 0.5, 0.75]          # annealing: increments start once n_model_updates >= warmup (VILoss.py:501-511)
    eng = model.engine
    for i, bt in enumerate(batches):
        stats = trainer.train([_Batch(bt, "cuda")], 1, None)
        ws = eng.workspace(B, S, T - 1)
        eps = ws.eps.view().cpu().clone()
        img = table[bt["indices"]]
        r, Lo, g = O.step_grads(po, c, bt["src"], bt["src_len"], bt["tgt"], img, eps, kl_mult=mults[i])
        assert abs(stats.nmt_loss - float(Lo["nll"])) <= 3e-5 * abs(float(Lo["nll"]))
        assert abs(stats.td_kl_before - float(Lo["kl_before"])) <= 3e-5 * abs(float(Lo["kl_before"]))
        assert abs(stats.td_kl_after - float(Lo["kl_after"])) <= 3e-5 * abs(float(Lo["kl_after"]))
        assert abs(stats.elbo_loss - float(Lo["elbo"])) <= 3e-5 * abs(float(Lo["elbo"]))
        assert stats.n_words == Lo["n_words"]
        assert abs(stats.td_kl_multiplier - mults[i]) < 1e-12
        po, _ = O.clip_and_adam(po, g, state, lr=0.002, max_grad_norm=5.0)
    # parameters after 3 updates (sign-level Adam differences excluded by comparing where the oracle moved clearly)
    for k in ("decoder.rnn.weight_hh_l0", "generator.0.bias", "inf_net_global.location.fc1.weight"):
        d = (model.state_dict()[k].cpu() - po[k]).abs().max().item()
        assert d < 5e-4, (k, d)
    # validation path (eval: z = mu) runs and produces finite statistics
    vs = trainer.validate([_Batch(batches[0], "cuda")])
    assert np.isfinite(vs.ppl()) and vs.n_words > 0
    # ---- checkpoint round trip ----------------------------------------------------------------------------------
    fname = trainer.drop_checkpoint(opt, 3, fields, vs)
    ck = torch.load(fname, map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == ["epoch", "generator", "model", "opt", "optim", "vocab"]
    assert sorted(ck["generator"].keys()) == ["0.bias", "0.weight"]
    assert not any("generator" in k for k in ck["model"])
    assert type(ck["optim"]).__module__ == "onmt.Optim"
    sd_opt = ck["optim"].optimizer.state_dict()
    assert len(sd_opt["state"]) == len(eng.grads) and int(float(next(iter(sd_opt["state"].values()))["step"])) == 3
    model2 = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, ck)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), model2.state_dict()[k].cpu()), k
    assert isinstance(ck["optim"].optimizer, torch.optim.Adam) and len(ck["optim"].params) == len(list(model.parameters()))
    optim2 = ck["optim"]
    optim2.optimizer.load_state_dict(ck["optim"].optimizer.state_dict())      # train_mm_vi_model1.py:434-438
    optim2.set_parameters(model2.parameters())
    # AS EXECUTED by the reference, set_parameters builds a NEW Adam (Optim.py:56-70): moments and step counter restart,
    # lr / _step / decay flags survive
    assert model2.engine.step_count == 0 and not model2.engine.flat_m.any() and not model2.engine.flat_v.any()
    assert optim2._step == 3 and optim2.lr == optim.lr
    # the opt-in extension carries the moments over (matched to the arena by shape + value, not by position)
    ck2 = torch.load(fname, map_location="cpu", weights_only=False)
    model3 = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, ck2)
    optim3 = ck2["optim"]
    optim3.resume_adam_state = True
    optim3.set_parameters(model3.parameters())
    assert model3.engine.step_count == 3
    assert torch.equal(model3.engine.flat_m.cpu(), eng.flat_m.cpu()) and torch.equal(model3.engine.flat_v.cpu(), eng.flat_v.cpu())


def test_conditional_model_through_the_onmt_surface(tmp_path):
    """--conditional driven through make_vi_model_mmt / TrainerMultimodal: two updates against the oracle,
    the prior's parameters in attns["p_latent"], and the shared-embedding alias in the state dict"""
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    c = O.Cfg(vs=43, vt=47, emb=16, hid=32, z=8, img=2048, layers=1, brnn=True, conditional=True)
    p = O.init_params(c, seed=4)
    B, S, T = 7, 6, 8
    batches = [O.synth_batch(c, B, S, T, n_img=20, seed=200 + i, fixed_len=False) for i in range(2)]
    table = batches[0]["table"]
    fields = {"src": types.SimpleNamespace(vocab=_Vocab(c.vs, False)), "tgt": types.SimpleNamespace(vocab=_Vocab(c.vt, True))}
    opt = _opt(c, tmp_path)
    opt.conditional = True
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, None)
    assert model.conditional and hasattr(model, "gen_net_global") and hasattr(model, "encoder_tgt")
    assert sorted(model.state_dict().keys()) == sorted(O.param_shapes(c).keys())
    alias = "encoder_tgt.embeddings.make_embedding.emb_luts.0.weight"
    assert alias in model.engine.state_dict()
    model.load_state_dict({k: v for k, v in p.items()})
    train_loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    trainer = onmt.TrainerMultimodal(model, train_loss, train_loss, optim, 0, 32, "text", "sents", 1, train_img_feats=table.numpy(),
                                     valid_img_feats=table.numpy(), multimodal_model_type="vi-model1", model_opt=opt, fields=fields)
    po, state = {k: v.clone() for k, v in p.items()}, {}
    eng = model.engine
    for bt in batches:
        stats = trainer.train([_Batch(bt, "cuda")], 1, None)
        ws = eng.workspace(B, S, T - 1)
        eps = ws.eps.view().cpu().clone()
        r, Lo, g = O.step_grads(po, c, bt["src"], bt["src_len"], bt["tgt"], table[bt["indices"]], eps, tgt_len=bt["tgt_len"])
        assert abs(stats.elbo_loss - float(Lo["elbo"])) <= 3e-5 * abs(float(Lo["elbo"]))
        assert abs(stats.td_kl_before - float(Lo["kl_before"])) <= 5e-5 * abs(float(Lo["kl_before"]))
        po, _ = O.clip_and_adam(po, g, state, lr=0.002, max_grad_norm=5.0)
    for k in ("gen_net_global.location.fc1.weight", "encoder_tgt.rnn.weight_hh_l0_reverse", "decoder.rnn.weight_hh_l0"):
        d = (model.state_dict()[k].cpu() - po[k]).abs().max().item()
        assert d < 5e-4, (k, d)
    vs = trainer.validate([_Batch(batches[0], "cuda")])
    assert np.isfinite(vs.ppl())


def test_two_layer_model_one_epoch_through_the_trainer_loop(tmp_path):
    """The scripts' kind of model (two uni-directional layers) through ONE call of TrainerMultimodal.train over four batches: inside the
    loop the engine holds the side-stream half of every update back until the next batch's forward (Engine.bg_after_head, the default for
    two or more layers; TrainerMultimodal._train_loop sets `hold_back`).  Four updates against the oracle's loop
    (onmt/TrainerMultimodal.py:679-711, onmt/Optim.py:78-96); every batch has a shape of its own, so that each workspace still holds the
    sample eps of ITS step afterwards."""
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    c = O.Cfg(vs=43, vt=47, emb=16, hid=32, z=8, img=2048, layers=2, brnn=False)
    p = O.init_params(c, seed=5)
    B = 7
    shapes = [(6, 8), (7, 9), (5, 7), (8, 10)]
    batches = [O.synth_batch(c, B, S, T, n_img=20, seed=300 + i, fixed_len=False) for i, (S, T) in enumerate(shapes)]
    table = batches[0]["table"]
    fields = {"src": types.SimpleNamespace(vocab=_Vocab(c.vs, False)), "tgt": types.SimpleNamespace(vocab=_Vocab(c.vt, True))}
    opt = _opt(c, tmp_path)
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, None)
    model.load_state_dict({k: v for k, v in p.items()})
    eng = model.engine
    import os
    assert eng.bg_after_head and eng.hold_back == (os.environ.get("VMMT_HOLD_BACK", "0") == "1")     # (off until a loop's owner sets it)
    train_loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    valid_loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    trainer = onmt.TrainerMultimodal(model, train_loss, valid_loss, optim, 0, 32, "text", "sents", 1, train_img_feats=table.numpy(),
                                     valid_img_feats=table.numpy(), multimodal_model_type="vi-model1", model_opt=opt, fields=fields)
    held = []
    real_step = eng.optim_step

    def spy(*a, **k):
        out = real_step(*a, **k)
        held.append(bool(eng.hold_back) and bool(eng._pending_bg))
        return out
    eng.optim_step = spy
    stats = trainer.train([_Batch(bt, "cuda") for bt in batches], 1, None)
    eng.optim_step = real_step
    assert held == [True] * 4                                   # every update of the loop left its side-stream half for the next forward
    assert not eng.hold_back and eng._pending_bg is None          # ... and the loop's end issued the last one
    # the oracle's loop with each step's own sample
    po, state, nll, nw = {k: v.clone() for k, v in p.items()}, {}, 0.0, 0
    for bt, (S, T) in zip(batches, shapes):
        eps = eng.workspace(B, S, T - 1).eps.view().cpu().clone()
        r, Lo, g = O.step_grads(po, c, bt["src"], bt["src_len"], bt["tgt"], table[bt["indices"]], eps)
        nll += float(Lo["nll"])
        nw += Lo["n_words"]
        po, _ = O.clip_and_adam(po, g, state, lr=0.002, max_grad_norm=5.0)
    assert stats.n_words == nw and abs(stats.nmt_loss - nll) <= 5e-5 * abs(nll)
    sd = model.state_dict()
    worst = max((sd[k].cpu() - po[k]).abs().max().item() for k in eng.grads
                if not k.startswith(("inf_net_image.location.fc1", "inf_net_image.gate_affine_transform")))
    assert worst < 5e-4, worst                                  # (four updates of lr 0.002: a systematic error is ~ 8e-3)
    assert eng.step_count == 4
