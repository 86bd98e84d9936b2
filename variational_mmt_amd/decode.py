"""Step-wise decoding with beam size 1 on the MI355X kernels (first half): the loop of
TranslatorMultimodalVI.translate_batch (onmt/translate/TranslatorMultimodalVI.py:114-200) -- encoder, z = E[q(z|x)] (fixed
prior) or E[p(z|x)] (conditional, :128-131), then one target position at a time: embedding, LSTM step(s), global attention,
generator, arg-max fed back -- issued through the same C-ABI as the training step.  Beam search proper (Beam.py) is not built."""
import ctypes as C

import torch

from . import _lib as L
from .engine import Buf, KPAD, PAD, _ru


def greedy_decode(eng, src, src_len, max_len=50, bos=2):
    """src [S,B] int64, src_len [B] (sorted descending).  Returns (tokens [max_len,B] int64, log-probs [max_len,B] f32) on the
    device; every sentence runs max_len steps (cut at the first </s> on the host).  No host synchronisation inside."""
    d, lib, dt, dev = eng.d, eng.lib, eng.dt, eng.dev
    S, B = int(src.shape[0]), int(src.shape[1])
    H, E, Z, V, Lyr = d.hid, d.emb, d.z, d.vt, d.layers
    # encoder + latent mean + z W_z^T: the evaluation-mode forward plan on a dummy 2-token target (its decoder step is ignored)
    dummy = torch.tensor([[bos] * B, [3] * B], dtype=torch.int64)
    tab = getattr(eng, "img_table", None)
    if tab is None:
        tab = torch.zeros(1, d.img, dtype=torch.float32, device=dev)        # the image row only feeds the training loss
    ws = eng.forward(src, src_len, dummy, torch.zeros(B, dtype=torch.int64), training=False, table=tab,
                     tgt_len=torch.full((B,), 2, dtype=torch.int64) if d.conditional else None)
    st = eng.stream()
    T, f32 = eng.T, torch.float32
    key = ("decode", B, S, max_len)
    bufs = eng.ws.get(key)
    if bufs is None:
        nb = lambda r, c, t=T: Buf(r, c, t, dev)
        bufs = dict(X=nb(B, E), gx=[nb(B, 4 * H, f32) for _ in range(Lyr)], gates=nb(B, 4 * H),
                    c=[nb(B, H, f32) for _ in range(Lyr)], h=[[nb(B, H), nb(B, H)] for _ in range(Lyr - 1)],
                    cat=[nb(B, 2 * H), nb(B, 2 * H)], Q=nb(B, H), AH=nb(B, H),
                    probs=torch.zeros(B * S, dtype=f32, device=dev),
                    tokens=torch.zeros(max_len + 1, B, dtype=torch.int64, device=dev),
                    vmax=torch.zeros(max_len, B, dtype=f32, device=dev), lse=torch.zeros(max_len, B, dtype=f32, device=dev),
                    npart=lib.vmmt_gen_npart(V))
        n = bufs["npart"] * B
        bufs.update(pm=torch.zeros(n, dtype=f32, device=dev), ps=torch.zeros(n, dtype=f32, device=dev),
                    pi=torch.zeros(n, dtype=torch.int32, device=dev), tl=torch.zeros(B, dtype=f32, device=dev),
                    nll=torch.zeros(B, dtype=f32, device=dev), stats=torch.zeros(L.STAT_COUNT, dtype=f32, device=dev))
        eng.ws[key] = bufs
    b = bufs
    tokens = b["tokens"]
    tokens[0].fill_(bos)

    def gemm(layout, A, lda, Bp, ldb, Cp, ldc, M, N, K, **kw):
        a = L.GemmArgs(dt, layout, A, lda, Bp, ldb, Cp, ldc, M, N, _ru(K, KPAD), 0, 0, kw.get("addend"), kw.get("ld_add", 0),
                       kw.get("add_rows", 0), 0, kw.get("act", L.ACT_NONE), kw.get("out_f32", 0), 0, 1.0, None, PAD, 0, 0)
        L.check(lib.vmmt_gemm(C.byref(a), st), "vmmt_gemm")

    # decoder state: h0 / c0 = encoder final states (Models.py:1158-1165); top layer's h lives in the right half of `cat`
    for l in range(Lyr):
        b["c"][l].view().copy_(ws.cn[l].view())
        (b["cat"][0].view()[:, H:] if l == Lyr - 1 else b["h"][l][0].view()).copy_(ws.hn[l].view())
    ctx = ws.enc_out[Lyr - 1]
    wa, wo, wg, we = eng.sh["wa"], eng.sh["wo"], eng.sh["wg"], eng.sh["dec_wih_l0_e"]
    esz = eng.tsz
    for t in range(max_len):
        cur, nxt = t & 1, (t + 1) & 1
        L.check(lib.vmmt_gather_rows(dt, eng.pp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                                     tokens.data_ptr() + 8 * t * B, b["X"].p(), b["X"].ld, B, E, st), "vmmt_gather_rows")
        x, xcols = b["X"], E
        for l in range(Lyr):
            top = l == Lyr - 1
            if l == 0:
                gemm(L.GEMM_NT, x.p(), x.ld, we.p(), we.ld, b["gx"][l].p(), b["gx"][l].ld, B, 4 * H, xcols, out_f32=1)
            else:
                wi, bs = eng.sh["dec_wih_l%d" % l], eng.sh["dec_b_l%d" % l]
                gemm(L.GEMM_NT, x.p(), x.ld, wi.p(), wi.ld, b["gx"][l].p(), b["gx"][l].ld, B, 4 * H, xcols, addend=bs.p(),
                     ld_add=bs.ld, add_rows=1, out_f32=1)
            arr = (L.LstmDirFwd * 2)()
            a = arr[0]
            hp = (b["cat"][cur], H) if top else (b["h"][l][cur], 0)
            ho = (b["cat"][nxt], H) if top else (b["h"][l][nxt], 0)
            whh = eng.sh["dec_whh_l%d" % l]
            a.h_prev, a.ld_hprev = hp[0].p(0, hp[1]), hp[0].ld
            a.c_prev, a.ld_cprev = b["c"][l].p(), b["c"][l].ld
            a.w_hh, a.ld_w = whh.p(), whh.ld
            a.gx, a.ld_gx = b["gx"][l].p(), b["gx"][l].ld
            if l == 0:
                a.gx2, a.ld_gx2 = ws.zx.p(), ws.zx.ld          # z W_z^T + b_ih + b_hh, constant over the sentence
            a.gates, a.ld_gates = b["gates"].p(), b["gates"].ld
            a.c_out, a.ld_c = b["c"][l].p(), b["c"][l].ld      # in place: a lane reads its own c_prev before it writes
            a.h_out, a.ld_h = ho[0].p(0, ho[1]), ho[0].ld
            a.t, a.capture = t, 0
            L.check(lib.vmmt_lstm_step_fwd(dt, 1, arr, None, B, H, st), "vmmt_lstm_step_fwd")
            x, xcols = ho[0], H
            xoff = ho[1]
        cat = b["cat"][nxt]
        gemm(L.GEMM_NT, cat.p(0, H), cat.ld, wa.p(), wa.ld, b["Q"].p(), b["Q"].ld, B, H, H)
        L.check(lib.vmmt_attn_fwd(dt, b["Q"].p(), b["Q"].ld, ctx.p(), ctx.ld, ws.src_len.data_ptr(), cat.p(), cat.ld,
                                  b["probs"].data_ptr(), 1, B, S, H, st), "vmmt_attn_fwd")
        gemm(L.GEMM_NT, cat.p(), cat.ld, wo.p(), wo.ld, b["AH"].p(), b["AH"].ld, B, H, 2 * H, act=L.ACT_TANH)
        L.check(lib.vmmt_gen_loss_fwd(dt, wg.p(), wg.ld, eng.pp("generator.0.bias"), b["AH"].p(), b["AH"].ld,
                                      tokens.data_ptr() + 8 * t * B, B, V, _ru(H, KPAD), PAD, b["pm"].data_ptr(), b["ps"].data_ptr(),
                                      b["pi"].data_ptr(), b["tl"].data_ptr(), b["lse"].data_ptr() + 4 * t * B, b["nll"].data_ptr(),
                                      b["stats"].data_ptr(), st), "vmmt_gen_loss_fwd")
        L.check(lib.vmmt_gen_argmax(b["pm"].data_ptr(), b["pi"].data_ptr(), B, b["npart"], tokens.data_ptr() + 8 * (t + 1) * B,
                                    b["vmax"].data_ptr() + 4 * t * B, st), "vmmt_gen_argmax")
    return tokens[1:], b["vmax"] - b["lse"]
