"""Step-wise decoding on the MI355X kernels: the loop of TranslatorMultimodalVI.translate_batch
(onmt/translate/TranslatorMultimodalVI.py:114-216) -- encoder, z = E[q(z|x)] (fixed prior) or E[p(z|x)] (conditional,
:128-131), then one target position at a time: embedding, LSTM step(s), global attention, generator -- issued through the
same C-ABI as the training step.  `greedy_decode`: beam size 1, arg-max fed back.  `beam_decode`: beam search proper; the
device runs Beam.advance (vmmt_beam_advance) and the beam re-ordering of the decoder state (vmmt_rows_select) for all
sentences of a batch with no host synchronisation inside a block of positions, and records every position's beam
(scores, parents, tokens, attention); `onmt/translate/Beam.py` (host mirror) replays the records into the reference's
bookkeeping (finished list, stopping rule, n-best extraction).

A position is ONE launch sequence with fixed arguments -- tokens ping through a pair of fixed buffers, the per-position records go
through staging buffers into history[counter] with the counter on the device (vmmt_history_append) -- issued launch by launch: a position
is bound by the dependent-kernel turnaround on the GPU, not by the host (replaying it as a captured hipGraph was built in round 3,
bit-identical and measured slower -- LABNOTES -- and removed in round 5).  What did pay was the two kernels the profile showed
(tools/decode_profile.py): Beam.advance as (row x vocabulary chunk) workgroups instead of one
per sentence (216 -> 39 us per position) and the arg-max fold as a wave per token (74 -> 5 us)."""
import ctypes as C

import torch

from . import _lib as L
from .engine import Buf, KPAD, PAD, _ru


class _Stepper(object):
    """Buffers and launches of one decoder position for R rows (R = B for arg-max decoding, K*B for beam search).
    State is double-buffered: a position reads set 0 (h of the lower layers, c, right half of cat[0]) and writes set 1."""

    def __init__(self, eng, R, S, ctx, ctx_ld, src_len, zx, zx_ld):
        self.eng, self.R, self.S = eng, R, S
        d, dev, T, f32 = eng.d, eng.dev, eng.T, torch.float32
        H, E, Lyr = d.hid, d.emb, d.layers
        Hp = d.hp                       # hidden size as computed (engine.Dims.hp): gate g at column g * Hp, [c ; h] halves at 0 / Hp
        nb = lambda r, c, t=T: Buf(r, c, t, dev)
        self.X = nb(R, E)
        self.gx = [nb(R, 4 * Hp, f32) for _ in range(Lyr)]
        self.gates = nb(R, 4 * Hp)
        self.c = [[nb(R, H, f32), nb(R, H, f32)] for _ in range(Lyr)]
        self.h = [[nb(R, H), nb(R, H)] for _ in range(Lyr - 1)]
        self.cat = [nb(R, 2 * Hp), nb(R, 2 * Hp)]
        self.Q, self.AH = nb(R, H), nb(R, H)
        self.probs = torch.zeros(R * S, dtype=f32, device=dev)
        self.ctx, self.ctx_ld, self.src_len, self.zx, self.zx_ld = ctx, ctx_ld, src_len, zx, zx_ld

    def gemm(self, layout, A, lda, Bp, ldb, Cp, ldc, M, N, K, **kw):
        e = self.eng
        a = L.GemmArgs(e.dt, layout, A, lda, Bp, ldb, Cp, ldc, M, N, _ru(K, KPAD), 0, 0, kw.get("addend"), kw.get("ld_add", 0),
                       kw.get("add_rows", 0), 0, kw.get("act", L.ACT_NONE), kw.get("out_f32", 0), 0, 1.0, None, PAD, 0, 0)
        L.check(e.lib.vmmt_gemm(C.byref(a), e.stream()), "vmmt_gemm")

    def h_buf(self, l, which):
        """(buffer, column offset) of layer l's hidden state in set `which`"""
        return (self.cat[which], self.eng.d.hp) if l == self.eng.d.layers - 1 else (self.h[l][which], 0)

    def step(self, tok_ptr, t):
        """tokens at `tok_ptr` (int64 [R]) -> AH = tanh(W_o [c ; h]) of this position, attention in self.probs; state 0 -> 1"""
        e = self.eng
        d, lib, dt, st, R = e.d, e.lib, e.dt, e.stream(), self.R
        H, E, Lyr, Hp = d.hid, d.emb, d.layers, d.hp
        L.check(lib.vmmt_gather_rows(dt, e.pp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E, tok_ptr,
                                     self.X.p(), self.X.ld, R, E, st), "vmmt_gather_rows")
        x, xoff, xcols = self.X, 0, E
        for l in range(Lyr):
            gx = self.gx[l]
            if l == 0:
                we = e.sh["dec_wih_l0_e"]
                self.gemm(L.GEMM_NT, x.p(0, xoff), x.ld, we.p(), we.ld, gx.p(), gx.ld, R, 4 * Hp, xcols, out_f32=1)
            else:
                wi, bs = e.sh["dec_wih_l%d" % l], e.sh["dec_b_l%d" % l]
                self.gemm(L.GEMM_NT, x.p(0, xoff), x.ld, wi.p(), wi.ld, gx.p(), gx.ld, R, 4 * Hp, xcols, addend=bs.p(), ld_add=bs.ld,
                          add_rows=1, out_f32=1)
            arr = (L.LstmDirFwd * 2)()
            a = arr[0]
            hp, ho = self.h_buf(l, 0), self.h_buf(l, 1)
            whh = e.sh["dec_whh_l%d" % l]
            a.h_prev, a.ld_hprev = hp[0].p(0, hp[1]), hp[0].ld
            a.c_prev, a.ld_cprev = self.c[l][0].p(), self.c[l][0].ld
            a.w_hh, a.ld_w = whh.p(), whh.ld
            a.gx, a.ld_gx = gx.p(), gx.ld
            if l == 0:
                a.gx2, a.ld_gx2 = self.zx, self.zx_ld            # z W_z^T + b_ih + b_hh, constant over the sentence
            a.gates, a.ld_gates = self.gates.p(), self.gates.ld
            a.c_out, a.ld_c = self.c[l][1].p(), self.c[l][1].ld
            a.h_out, a.ld_h = ho[0].p(0, ho[1]), ho[0].ld
            a.t, a.capture = t, 0
            L.check(lib.vmmt_lstm_step_fwd(dt, 1, arr, None, R, Hp, st), "vmmt_lstm_step_fwd")
            x, xoff, xcols = ho[0], ho[1], H
        cat = self.cat[1]
        wa, wo = e.sh["wa"], e.sh["wo"]
        self.gemm(L.GEMM_NT, cat.p(0, Hp), cat.ld, wa.p(), wa.ld, self.Q.p(), self.Q.ld, R, H, H)
        L.check(lib.vmmt_attn_fwd(dt, self.Q.p(), self.Q.ld, self.ctx, self.ctx_ld, self.src_len.data_ptr(), cat.p(), cat.ld,
                                  self.probs.data_ptr(), 1, R, self.S, Hp, st), "vmmt_attn_fwd")
        self.gemm(L.GEMM_NT, cat.p(), cat.ld, wo.p(), wo.ld, self.AH.p(), self.AH.ld, R, H, 2 * Hp, act=L.ACT_TANH)

    def carry(self, rows_ptr=None):
        """state set 1 -> set 0 for the next position; `rows_ptr` (int64 [R]) selects each row's parent (beam re-ordering,
        RNNDecoderState.beam_update, onmt/Models.py:589-594), None = identity"""
        e = self.eng
        H, Lyr, st = e.d.hid, e.d.layers, e.stream()
        for l in range(Lyr):
            pairs = [(self.h_buf(l, 1), self.h_buf(l, 0), e.tsz), ((self.c[l][1], 0), (self.c[l][0], 0), 4)]
            for (sb, so), (db, do), esz in pairs:
                if rows_ptr is None:
                    db.t[:self.R, do:do + H].copy_(sb.t[:self.R, so:so + H])
                else:
                    L.check(e.lib.vmmt_rows_select(sb.p(0, so), sb.ld * esz, rows_ptr, db.p(0, do), db.ld * esz, self.R, H * esz, st),
                            "vmmt_rows_select")


def _encode(eng, src, src_len, bos):
    """encoder + latent mean + z W_z^T: the evaluation-mode forward plan on a dummy 2-token target (its decoder step is ignored)"""
    d, dev = eng.d, eng.dev
    B = int(src.shape[1])
    dummy = torch.tensor([[bos] * B, [3] * B], dtype=torch.int64, device="cpu")
    tab = getattr(eng, "img_table", None)
    if tab is None:
        tab = torch.zeros(1, d.img, dtype=torch.float32, device=dev)        # the image row only feeds the training loss
    return eng.forward(src, src_len, dummy, torch.zeros(B, dtype=torch.int64, device="cpu"), training=False, table=tab,
                       tgt_len=torch.full((B,), 2, dtype=torch.int64, device="cpu") if d.conditional else None)


def _hist(eng, segs, counter, limit):
    """vmmt_history_append: [(src tensor, history tensor [limit][...])] -> history[counter] <- src; counter += 1"""
    arr = (L.HistSeg * len(segs))()
    for a, (src, hist) in zip(arr, segs):
        nbytes = src.numel() * src.element_size()
        assert hist[0].numel() * hist.element_size() == nbytes and hist.shape[0] >= limit and src.is_contiguous() and hist.is_contiguous()
        a.src, a.dst, a.bytes, a.stride_bytes = src.data_ptr(), hist.data_ptr(), nbytes, nbytes
    L.check(eng.lib.vmmt_history_append(arr, len(segs), counter.data_ptr(), limit, 1, eng.stream()), "vmmt_history_append")


def greedy_decode(eng, src, src_len, max_len=50, bos=2, eos=None, check_every=16):
    """src [S,B] int64, src_len [B] (sorted descending).  Returns (tokens [n,B] int64, log-probs [n,B] f32) on the device, n <=
    max_len positions (the host cuts every sentence at its first </s>).  eos=None: all max_len positions, no host synchronisation
    inside; with `eos` the loop looks every `check_every` positions whether every sentence has produced </s> (one small
    device-to-host copy) and stops there -- the reference's validation translations run with max_length 100 for sentences of ~15."""
    d, lib, dt, dev = eng.d, eng.lib, eng.dt, eng.dev
    S, B = int(src.shape[0]), int(src.shape[1])
    H, V, Lyr = d.hid, d.vt, d.layers
    ws = _encode(eng, src, src_len, bos)
    f32 = torch.float32
    key = ("decode", B, S, max_len)
    b = eng.ws.get(key)
    enc = ws.enc_out[Lyr - 1]
    if b is None:
        b = dict(tokens=torch.zeros(max_len, B, dtype=torch.int64, device=dev), tok=torch.zeros(2, B, dtype=torch.int64, device=dev),
                 vmax=torch.zeros(max_len, B, dtype=f32, device=dev), lse=torch.zeros(max_len, B, dtype=f32, device=dev),
                 st_vmax=torch.zeros(B, dtype=f32, device=dev), st_lse=torch.zeros(B, dtype=f32, device=dev),
                 counter=torch.zeros(1, dtype=torch.int32, device=dev), npart=lib.vmmt_gen_npart(V),
                 # the source-side operands in buffers of this workspace (a position runs on fixed addresses)
                 ctx=Buf(S * B, enc.t.shape[1], eng.T, dev, ld=enc.ld), zx=Buf(B, ws.zx.t.shape[1], f32, dev, ld=ws.zx.ld),
                 src_len=torch.zeros(B, dtype=torch.int64, device=dev))
        n = b["npart"] * B
        b.update(pm=torch.zeros(n, dtype=f32, device=dev), ps=torch.zeros(n, dtype=f32, device=dev),
                 pi=torch.zeros(n, dtype=torch.int32, device=dev), tl=torch.zeros(B, dtype=f32, device=dev),
                 nll=torch.zeros(B, dtype=f32, device=dev), stats=torch.zeros(L.STAT_COUNT, dtype=f32, device=dev))
        b["stepper"] = _Stepper(eng, B, S, b["ctx"].p(), b["ctx"].ld, b["src_len"], b["zx"].p(), b["zx"].ld)
        eng.ws[key] = b
    sp, tok = b["stepper"], b["tok"]
    wg = eng.sh["wg"]

    def position():
        st = eng.stream()
        sp.step(tok[0].data_ptr(), 0)
        L.check(lib.vmmt_gen_loss_fwd(dt, wg.p(), wg.ld, eng.pp("generator.0.bias"), sp.AH.p(), sp.AH.ld, tok[0].data_ptr(), B, V,
                                      _ru(H, KPAD), PAD, b["pm"].data_ptr(), b["ps"].data_ptr(), b["pi"].data_ptr(), b["tl"].data_ptr(),
                                      b["st_lse"].data_ptr(), b["nll"].data_ptr(), b["stats"].data_ptr(), st), "vmmt_gen_loss_fwd")
        L.check(lib.vmmt_gen_argmax(b["pm"].data_ptr(), b["pi"].data_ptr(), B, b["npart"], tok[1].data_ptr(), b["st_vmax"].data_ptr(), st),
                "vmmt_gen_argmax")
        _hist(eng, [(tok[1], b["tokens"]), (b["st_lse"], b["lse"]), (b["st_vmax"], b["vmax"])], b["counter"], max_len)
        sp.carry()
        tok[0].copy_(tok[1])

    def init():
        b["ctx"].t.copy_(enc.t[:b["ctx"].t.shape[0]])
        b["zx"].t[:B].copy_(ws.zx.t[:B])
        b["src_len"].copy_(ws.src_len)
        tok[0].fill_(bos)
        b["counter"].zero_()
        # decoder state: h0 / c0 = encoder final states (Models.py:1158-1165)
        for l in range(Lyr):
            sp.c[l][0].view().copy_(ws.cn[l].view())
            hb, ho = sp.h_buf(l, 0)
            hb.t[:B, ho:ho + H].copy_(ws.hn[l].view())

    init()
    n = max_len
    for t in range(max_len):
        position()
        if eos is not None and (t + 1) % check_every == 0 and t + 1 < max_len and bool((b["tokens"][:t + 1] == eos).any(0).all()):
            n = t + 1
            break
    return b["tokens"][:n], (b["vmax"] - b["lse"])[:n]


def beam_decode(eng, src, src_len, beam_size, max_len=100, min_length=0, bos=2, eos=3, pad=PAD, stop=None, check_every=8):
    """Beam search over a batch: src [S,B] int64, src_len [B] (sorted descending).  Runs positions 0 .. max_len-1 on the device
    (rows k*B + b) and returns the per-position records as HOST tensors:
        scores [n,B,K] f32, prev [n,B,K] int32, next [n,B,K] int64, attn [n,K*B,S] f32       (n = positions run)
    `stop(records) -> bool` is consulted every `check_every` positions (one device-to-host copy each time) so that the loop
    ends once every sentence's beam is done (Beam.done, evaluated by the host mirror)."""
    d, lib, dt, dev = eng.d, eng.lib, eng.dt, eng.dev
    S, B, K = int(src.shape[0]), int(src.shape[1]), int(beam_size)
    if not 1 <= K <= 16:
        raise ValueError("beam_size %d outside 1..16" % K)
    H, V, Lyr = d.hid, d.vt, d.layers
    R = K * B
    ws = _encode(eng, src, src_len, bos)
    f32, i64 = torch.float32, torch.int64
    key = ("beam", B, S, K, max_len, int(eos))
    b = eng.ws.get(key)
    if b is None:
        b = dict(ctx=Buf(S * R, H, eng.T, dev), zx=Buf(R, 4 * d.hp, f32, dev), src_len=torch.zeros(R, dtype=i64, device=dev),
                 logits=Buf(R, V, f32, dev), tok=torch.zeros(2, R, dtype=i64, device=dev), sel=torch.zeros(R, dtype=i64, device=dev),
                 scores=torch.zeros(B, K, dtype=f32, device=dev), h_score=torch.zeros(max_len, B, K, dtype=f32, device=dev),
                 h_prev=torch.zeros(max_len, B, K, dtype=torch.int32, device=dev),
                 h_next=torch.zeros(max_len, B, K, dtype=i64, device=dev), h_attn=torch.zeros(max_len, R, S, dtype=f32, device=dev),
                 st_score=torch.zeros(B, K, dtype=f32, device=dev), st_prev=torch.zeros(B, K, dtype=torch.int32, device=dev),
                 st_next=torch.zeros(B, K, dtype=i64, device=dev), counter=torch.zeros(1, dtype=torch.int32, device=dev),
                 adv_ws=torch.zeros(int(lib.vmmt_beam_advance_ws_bytes(B, K, V)) // 4, dtype=f32, device=dev))
        b["stepper"] = _Stepper(eng, R, S, b["ctx"].p(), b["ctx"].ld, b["src_len"], b["zx"].p(), b["zx"].ld)
        eng.ws[key] = b
    sp, tok = b["stepper"], b["tok"]
    wg, lg = eng.sh["wg"], b["logits"]
    bias = eng.pp("generator.0.bias")

    def position(first=0, mask_eos=0):
        sp.step(tok[0].data_ptr(), 0)
        sp.gemm(L.GEMM_NT, sp.AH.p(), sp.AH.ld, wg.p(), wg.ld, lg.p(), lg.ld, R, V, H, addend=bias, ld_add=0, add_rows=1, out_f32=1)
        L.check(lib.vmmt_beam_advance(lg.p(), lg.ld, B, K, V, tok[0].data_ptr(), b["scores"].data_ptr(), first, mask_eos, eos,
                                      tok[1].data_ptr(), b["sel"].data_ptr(), b["st_score"].data_ptr(), b["st_prev"].data_ptr(),
                                      b["st_next"].data_ptr(), b["adv_ws"].data_ptr(), b["adv_ws"].numel() * 4, eng.stream()),
                "vmmt_beam_advance")
        _hist(eng, [(b["st_score"], b["h_score"]), (b["st_prev"], b["h_prev"]), (b["st_next"], b["h_next"]), (sp.probs, b["h_attn"])],
              b["counter"], max_len)
        sp.carry(b["sel"].data_ptr())
        tok[0].copy_(tok[1])

    def init():
        # (2) repeat the source-side objects beam_size times (TranslatorMultimodalVI.py:141-157): row k*B + b <- sentence b
        enc = ws.enc_out[Lyr - 1]
        b["ctx"].t[:S * R].view(S, K, B, -1).copy_(enc.t[:S * B].view(S, 1, B, -1).expand(S, K, B, enc.ld))
        b["zx"].t[:R].view(K, B, -1).copy_(ws.zx.t[:B].unsqueeze(0).expand(K, B, ws.zx.ld))
        b["src_len"].view(K, B).copy_(ws.src_len.view(1, B).expand(K, B))
        for l in range(Lyr):
            sp.c[l][0].t[:R].view(K, B, -1).copy_(ws.cn[l].t[:B].unsqueeze(0).expand(K, B, ws.cn[l].ld))
            hb, ho = sp.h_buf(l, 0)
            hb.t[:R, ho:ho + H].view(K, B, H).copy_(ws.hn[l].view().unsqueeze(0).expand(K, B, H))
        tok[0].fill_(pad)                         # Beam.__init__: next_ys[0] = [bos, pad, pad, ...] (Beam.py:34-36)
        tok[0, :B].fill_(bos)
        b["scores"].zero_()
        b["counter"].zero_()

    init()
    n = 0
    for t in range(max_len):
        first, mask_eos = int(t == 0), int(t + 1 < min_length)
        position(first, mask_eos)             # the first position scores beam 0 only; positions below min_length mask </s>
        n = t + 1
        if stop is not None and (n % check_every == 0 or n == max_len):
            rec = dict(scores=b["h_score"][:n].cpu(), prev=b["h_prev"][:n].cpu(), next=b["h_next"][:n].cpu(), attn=None)
            if stop(rec):
                break
    return dict(scores=b["h_score"][:n].cpu(), prev=b["h_prev"][:n].cpu(), next=b["h_next"][:n].cpu(),
                attn=b["h_attn"][:n].cpu().view(n, K, B, S), src_len=ws.src_len.cpu())
