"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by running the REAL reference
(imported from /root/reference through oracle/ref_harness.py) on seeded inputs.
Run in the build container only:   python oracle/make_golden.py

Each fixture holds: the reference model's state_dict, the inputs (src/tgt/lengths/image
rows/eps), forward intermediates + loss statistics from the reference's *monolithic* loss
path (as-executed H1 image term), parameter gradients from the reference's *sharded
training* path (`loss.div(B).backward()`, with shims s6/s7 => image-term semantic "A"),
and the parameters after one `onmt.Optim.step()` (clip 5 + Adam lr 0.002).
Fixtures are data (inputs + expected outputs) -- no reference source text is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness as RH          # noqa: E402
from oracle import vi1_oracle as O            # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (cfg kwargs, B, S, T, fixed_len)
    "tiny_uni_l2": (dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=2, brnn=False), 5, 7, 9, False),
    "tiny_bi_l1":  (dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True), 5, 7, 9, False),
    "tiny_bi_l2":  (dict(vs=53, vt=47, emb=20, hid=24, z=12, layers=2, brnn=True), 6, 9, 8, False),
    "small_fixed": (dict(vs=211, vt=307, emb=40, hid=64, z=32, layers=1, brnn=True), 8, 10, 11, True),
    # --conditional prior: p(z|x) = gen_net_global, q(z|x,y,v) = GlobalFullInferenceNetwork, encoder_tgt
    "cond_bi_l1":  (dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True, conditional=True), 5, 7, 9, False),
    "cond_uni_l2": (dict(vs=43, vt=39, emb=10, hid=12, z=6, layers=2, brnn=False, conditional=True), 6, 8, 7, False),
    # BASELINE.json config 1 at its REAL shape (run_translated_m30k_only.sh defaults: batch 40, 30k vocabularies, biLSTM 512,
    # z 256, emb 500): every tensor above 65 536 elements is formula-initialised / kept as a strided sample + sums, so the
    # fixture stays a few MB.  Reaches the bf16 *_fast kernels (H % 32 == 0, LDS-DMA GEMM loops) with reference-held values.
    "cfg1_shape":  (dict(vs=30000, vt=30000, emb=500, hid=512, z=256, layers=1, brnn=True), 40, 20, 21, False),
    # the run scripts AS WRITTEN (run_translated_m30k_only.sh:46-57 + opts.py defaults: -rnn_size 500 --z_latent_dim 500, word vectors
    # 500, 2-layer uni-directional LSTMs, batch 40): hidden sizes that are no multiple of anything the MFMA kernels tile -- the build
    # computes them padded to 512 (engine.Dims.hp) and must reproduce the reference exactly in the 500 real lanes
    "script_shape": (dict(vs=30000, vt=30000, emb=500, hid=500, z=500, layers=2, brnn=False), 40, 20, 21, False),
}


def run_case(name, ck, B, S, T, fixed):
    onmt, _ = RH.import_reference()
    c = O.Cfg(**ck)
    opt = RH.make_opt(src_word_vec_size=c.emb, tgt_word_vec_size=c.emb, rnn_size=c.hid, z_latent_dim=c.z,
                      enc_layers=c.layers, dec_layers=c.layers, encoder_type="brnn" if c.brnn else "rnn",
                      dropout=0.0, conditional=bool(c.conditional))
    model, fields = RH.build_model(opt, c.vs, c.vt, seed=0)
    model.train()
    for k, v in model.named_parameters():          # large tensors: formula values (not stored in the fixture)
        if v.numel() > O.BIG:
            v.data.copy_(O.formula_param(k, tuple(v.shape)))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bt = O.synth_batch(c, B, S, T, n_img=16, seed=4321 + len(name), fixed_len=fixed)
    img = bt["table"][bt["indices"]]
    src3, tgt3 = bt["src"].unsqueeze(2), bt["tgt"].unsqueeze(2)
    out = {}
    # ---- forward + monolithic loss (as-executed values) --------------------------------
    vloss = RH.make_loss(model, fields, opt, training=False)
    with RH.inject_eps(bt["eps"]):
        outputs, attns, _ = model(src3, tgt3, bt["src_len"], bt["tgt_len"], img.clone())
    batch = RH.Batch(bt["src"], bt["src_len"], bt["tgt"], bt["tgt_len"], bt["indices"])
    batch.tgt = batch.tgt[0]
    with torch.no_grad():
        stats = vloss.monolithic_compute_loss(batch, outputs, {k: v for k, v in attns.items()})
    out["f_out"] = outputs
    out["f_attn"] = attns["std"]
    with torch.no_grad():      # per-token NLL through the reference's own generator (ModelConstructor.py:583-585) and target shift
        scores = model.generator(outputs.reshape(-1, outputs.size(2)))
        yy = bt["tgt"][1:].reshape(-1)
        out["f_tok_nll"] = (-scores.gather(1, yy.unsqueeze(1)).squeeze(1) * (yy != 1).float()).view(T - 1, B)
    out["f_mu"], out["f_sigma"] = attns["z_latent"][0].params()
    out["f_z"] = attns["z0_sample"][0]
    if c.conditional:
        out["f_mu_p"], out["f_sigma_p"] = attns["p_latent"][0].params()
    # NB: monolithic loss normalised p_v / img in place (H1) -> re-run the image net for the raw mean
    with torch.no_grad():
        out["f_mu_v"] = model.inf_net_image(out["f_z"], None, None)[0].params()[0]
    for k in ("nmt_loss", "td_kl_before", "td_kl_after", "image_feats_loss", "image_feats_cos", "elbo_loss"):
        out["s_" + k] = torch.as_tensor(float(getattr(stats, k)))
    out["s_n_words"] = torch.as_tensor(int(stats.n_words))
    out["s_n_correct"] = torch.as_tensor(int(stats.n_correct))
    # encoder internals
    with torch.no_grad():
        enc_hidden, context = model.encoder(src3, bt["src_len"])
    out["f_context"], out["f_enc_h"], out["f_enc_c"] = context, enc_hidden[0], enc_hidden[1]
    # ---- sharded training loss + backward + one optimiser step -----------------------
    tloss = RH.make_loss(model, fields, opt, training=True)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    model.zero_grad()
    with RH.inject_eps(bt["eps"]), RH.training_shims():
        outputs, attns, _ = model(src3, tgt3, bt["src_len"], bt["tgt_len"], img.clone())
        batch = RH.Batch(bt["src"], bt["src_len"], bt["tgt"], bt["tgt_len"], bt["indices"])
        batch.tgt = batch.tgt[0]
        tstats = tloss.sharded_compute_loss(batch, outputs, attns, 0, T, 32, B)
    out["t_elbo"] = torch.as_tensor(float(tstats.elbo_loss))
    out["t_img_logprob_A"] = torch.as_tensor(float(tstats.image_feats_loss))
    for k, v in model.named_parameters():
        if v.grad is not None:
            out["g_" + k] = v.grad.detach().clone()
    optim.step()
    for k, v in model.state_dict().items():
        out["p1_" + k] = v.detach().clone()
    arrs = {"in_" + k: v.numpy() for k, v in bt.items()}
    arrs.update({"p0_" + k: v.numpy() for k, v in sd.items()})
    arrs.update({k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    arrs["cfg"] = np.array([c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, int(c.brnn), B, S, T, int(c.conditional)], dtype=np.int64)
    os.makedirs(OUT, exist_ok=True)
    # the image MLP's fc2 is 2048x2048 (x2, 33 MB): store fp16-exact-free? no -- keep fp32 but drop the dead
    # scale branch (never read by forward/loss; H6) to keep fixtures small
    arrs = {k: v for k, v in arrs.items() if "inf_net_image.scale" not in k}
    for k in list(arrs):
        if k[:3] in ("p0_", "p1_") or k[:2] == "g_":
            if arrs[k].size > O.BIG:
                t = torch.from_numpy(arrs.pop(k))
                if k[:3] == "p0_":
                    continue                       # regenerated by O.formula_param in the tests
                sub, s1, s2 = O.sample_big(t)
                arrs["big_" + k] = sub.numpy()
                arrs["bigsum_" + k] = np.array([float(s1), float(s2)])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(name, "elbo", float(out["s_elbo_loss"]), "nmt", float(out["s_nmt_loss"]), "kl", float(out["s_td_kl_before"]),
          "img", float(out["s_image_feats_loss"]), "words", int(out["s_n_words"]))


GREEDY = {
    # name: (cfg kwargs, B, S, max_len, weight scale): decoding with beam size 1 through the reference's OWN modules
    "greedy_bi_l1": (dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True), 6, 7, 10, 6.0),
    "greedy_cond_uni_l2": (dict(vs=43, vt=39, emb=10, hid=12, z=6, layers=2, brnn=False, conditional=True), 5, 8, 9, 6.0),
}


def run_greedy(name, ck, B, S, max_len, scale):
    """tokens of TranslatorMultimodalVI's step loop with beam size 1 (translate/TranslatorMultimodalVI.py:114-200), driven
    through the reference's encoder / latent network / decoder / generator modules"""
    onmt, _ = RH.import_reference()
    c = O.Cfg(**ck)
    opt = RH.make_opt(src_word_vec_size=c.emb, tgt_word_vec_size=c.emb, rnn_size=c.hid, z_latent_dim=c.z,
                      enc_layers=c.layers, dec_layers=c.layers, encoder_type="brnn" if c.brnn else "rnn",
                      dropout=0.0, conditional=bool(c.conditional))
    model, fields = RH.build_model(opt, c.vs, c.vt, seed=0)
    model.eval()
    for k, v in model.named_parameters():           # larger weights than the +-0.1 initialisation: varied arg-max paths
        if "inf_net_image" not in k:
            v.data.mul_(scale)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bt = O.synth_batch(c, B, S, 5, n_img=8, seed=77 + len(name), fixed_len=False)
    src3 = bt["src"].unsqueeze(2)
    toks, scores = [], []
    with torch.no_grad():
        enc_states, context = model.encoder(src3, bt["src_len"])
        net = model.gen_net_global if c.conditional else model.inf_net_global
        q0, _ = net(context.detach(), bt["src_len"])
        s0 = q0.mean()
        st = model.decoder.init_decoder_state(src3, context, enc_states)
        inp = torch.full((1, B, 1), 2, dtype=torch.int64)
        for _ in range(max_len):
            dec_out, st, _ = model.decoder(inp, context, st, z_sample=s0, image_features=None, context_lengths=bt["src_len"])
            out = model.generator.forward(dec_out.squeeze(0))
            s, nxt = out.max(1)
            toks.append(nxt); scores.append(s)
            inp = nxt.view(1, -1, 1)
    arrs = {"in_src": bt["src"].numpy(), "in_src_len": bt["src_len"].numpy()}
    arrs.update({"p0_" + k: v.numpy() for k, v in sd.items() if "inf_net_image.scale" not in k and v.numel() <= O.BIG})
    arrs["tokens"] = torch.stack(toks).numpy()
    arrs["scores"] = torch.stack(scores).numpy()
    arrs["cfg"] = np.array([c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, int(c.brnn), B, S, max_len, int(c.conditional)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(name, "tokens", arrs["tokens"][:, :3].T.tolist())


BEAM = {
    # name: (cfg kwargs, B, S, max_len, weight scale, </s> bias, beam size, n_best, alpha, beta, min_length): beam search through
    # the reference's OWN TranslatorMultimodalVI.translate_batch + Beam + GNMTGlobalScorer, one sentence at a time
    # (translate_mm_vi.py:80-82 forces batch size 1)
    "beam_bi_l1": (dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True), 6, 7, 12, 6.0, 0.7, 5, 3, 0.0, -0.0, 0),
    "beam_cond_uni_l2": (dict(vs=43, vt=39, emb=10, hid=12, z=6, layers=2, brnn=False, conditional=True), 5, 8, 10, 6.0, 0.7,
                         3, 1, 0.6, 0.2, 3),
    "beam_bi_l2_alpha": (dict(vs=53, vt=47, emb=20, hid=24, z=12, layers=2, brnn=True), 5, 9, 11, 6.0, 0.9, 4, 2, 1.0, -0.0, 0),
    "beam_bi_l1_k2": (dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True), 4, 6, 9, 5.0, 0.6, 2, 2, 0.0, -0.0, 0),
}


def run_beam(name, ck, B, S, max_len, scale, eos_bias, K, n_best, alpha, beta, min_length):
    import types
    onmt, _ = RH.import_reference()
    import onmt.translate
    c = O.Cfg(**ck)
    opt = RH.make_opt(src_word_vec_size=c.emb, tgt_word_vec_size=c.emb, rnn_size=c.hid, z_latent_dim=c.z,
                      enc_layers=c.layers, dec_layers=c.layers, encoder_type="brnn" if c.brnn else "rnn",
                      dropout=0.0, conditional=bool(c.conditional))
    model, fields = RH.build_model(opt, c.vs, c.vt, seed=0)
    model.eval()
    for k, v in model.named_parameters():
        if "inf_net_image" not in k:
            v.data.mul_(scale)
    model.generator[0].bias.data[3] += eos_bias          # so that </s> reaches the beams after a few positions
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bt = O.synth_batch(c, B, S, 5, n_img=8, seed=91 + len(name), fixed_len=False)
    scorer = onmt.translate.GNMTGlobalScorer(alpha, beta)
    tr = onmt.translate.TranslatorMultimodalVI(model, fields, beam_size=K, n_best=n_best, global_scorer=scorer,
                                               max_length=max_len, copy_attn=False, cuda=False, min_length=min_length,
                                               test_img_feats=bt["table"].numpy(), multimodal_model_type="vi-model1")
    beams = []
    from_beam = tr._from_beam
    tr._from_beam = lambda bm: (beams.extend(bm), from_beam(bm))[1]
    data = types.SimpleNamespace(data_type="text")
    arrs = {"in_src": bt["src"].numpy(), "in_src_len": bt["src_len"].numpy()}
    arrs.update({"p0_" + k: v.numpy() for k, v in sd.items() if "inf_net_image.scale" not in k and v.numel() <= O.BIG})
    pred = -np.ones((B, n_best, max_len), dtype=np.int64)
    plen = np.zeros((B, n_best), dtype=np.int64)
    score = np.zeros((B, n_best), dtype=np.float32)
    attn = np.zeros((B, n_best, max_len, S), dtype=np.float32)
    steps = np.zeros(B, dtype=np.int64)
    h_next = -np.ones((B, max_len + 1, K), dtype=np.int64)
    h_prev = -np.ones((B, max_len, K), dtype=np.int64)
    h_score = np.zeros((B, max_len, K), dtype=np.float32)
    import warnings
    warnings.simplefilter("ignore")
    with RH.beam_shims(), torch.no_grad():
        for b in range(B):
            n = int(bt["src_len"][b])
            batch = types.SimpleNamespace(src=(bt["src"][:n, b:b + 1], bt["src_len"][b:b + 1]), batch_size=1)
            ret = tr.translate_batch(batch, data, b)
            for i in range(n_best):
                hyp = [int(t) for t in ret["predictions"][0][i]]
                pred[b, i, :len(hyp)] = hyp
                plen[b, i] = len(hyp)
                score[b, i] = float(ret["scores"][0][i])
                a = ret["attention"][0][i]
                attn[b, i, :a.shape[0], :a.shape[1]] = a.numpy()
            bm = beams[b]
            steps[b] = len(bm.prev_ks)
            for t, y in enumerate(bm.next_ys):
                h_next[b, t] = y.numpy()
            for t, (pk, sc) in enumerate(zip(bm.prev_ks, bm.all_scores[1:] + [bm.scores])):
                h_prev[b, t] = pk.numpy()
                h_score[b, t] = sc.numpy()
    arrs.update(pred=pred, pred_len=plen, score=score, attention=attn, steps=steps, hist_next=h_next, hist_prev=h_prev,
                hist_score=h_score, beam=np.array([K, n_best, min_length], dtype=np.int64),
                scorer=np.array([alpha, beta], dtype=np.float64))
    arrs["cfg"] = np.array([c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, int(c.brnn), B, S, max_len, int(c.conditional)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(name, "steps", steps.tolist(), "best", [pred[b, 0, :plen[b, 0]].tolist() for b in range(B)], score[:, 0].tolist())


if __name__ == "__main__":
    only = sys.argv[1:]
    for n, a in BEAM.items():
        if only and n not in only:
            continue
        run_beam(n, *a)
    for n, (ck, B, S, ml, sc) in GREEDY.items():
        if only and n not in only:
            continue
        run_greedy(n, ck, B, S, ml, sc)
    for n, (ck, B, S, T, fx) in CASES.items():
        if only and n not in only:
            continue
        run_case(n, ck, B, S, T, fx)
