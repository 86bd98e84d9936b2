"""Buffers and plan pieces of the `--conditional` prior variant (mixin of workspace.Workspace)."""
import ctypes as C
import math
from os import environ as _os_env

import torch

from .. import _lib as L
from .layout import Buf, KPAD, PAD, _ru  # noqa: F401


class ConditionalPlans(object):
    def _cond_alloc(self):
        """buffers of the conditional branch: p(z|x), q(z|x,y,v) input [h_x ; h_y ; v], encoder_tgt (rows b*T + t: the
        reference runs it over the transposed target, so its recurrence walks the batch axis -- hazard H5)"""
        e, d = self.e, self.e.d
        T, dev = e.T, e.dev
        f32, i64 = torch.float32, torch.int64
        B, H, ht, E, Z, Lyr = self.B, d.hid, d.ht, d.emb, d.z, d.layers
        htp = d.htp
        Ht = 2 * htp                       # encoder_tgt's output as computed: [fwd : htp | bwd : htp] (Dims.htp)
        Tn = self.Tp + 1
        MT = B * Tn
        self.Tn, self.MT = Tn, MT
        nb = lambda r, c, dt=T, **kw: Buf(r, c, dt, dev, **kw)
        self.tgt_bt = torch.zeros(MT, dtype=i64, device=dev)
        self.tgt_len = torch.zeros(B, dtype=i64, device=dev)
        self.Yt = nb(MT, E)
        self.enct_gx = [nb(MT, 8 * htp, f32) for _ in range(Lyr)]
        self.enct_gates = [nb(MT, 8 * htp) for _ in range(Lyr)]
        self.enct_c = [nb(MT, Ht, f32) for _ in range(Lyr)]
        self.enct_out = [nb(MT, Ht) for _ in range(Lyr)]
        self.enct_mask = [nb(MT, Ht) if (d.dropout > 0 and l < Lyr - 1) else None for l in range(Lyr)]
        self.enct_xdrop = [nb(MT, Ht) if (d.dropout > 0 and l < Lyr - 1) else None for l in range(Lyr)]
        self.enct_hzero = nb(Tn, Ht)
        self.hq = nb(B, d.qin_p)
        self.mu_p = nb(B, Z, f32, ld=Z)
        self.sigma_p = nb(B, Z, f32, ld=Z)
        self.p_h1 = {br: nb(B, Z) for br in ("location", "scale")}
        self.p_dmu = nb(B, Z)
        self.p_dpre = nb(B, Z)
        self.p_dh1 = {br: nb(B, Z) for br in ("location", "scale")}
        self.dhbar_p = nb(B, H)
        self.dhy = nb(B, Ht)
        self.enct_dout = nb(MT, Ht)
        self.enct_dgates = [nb(MT, 8 * htp) for _ in range(Lyr)]
        self.enct_dcc = [nb(Tn, Ht, f32) for _ in range(Lyr)]
        self.enct_dx = [nb(MT, Ht) for _ in range(Lyr - 1)]

    def _cond_forward_aux(self, P, training):
        """aux stream: encoder_tgt over the transposed target (B recurrent steps with T rows each; Models.py:892-894)"""
        e, d, lib = self.e, self.e.d, self.e.lib
        B, S, H, E, Z, D, Lyr = self.B, self.S, d.hid, d.emb, d.z, d.img, d.layers
        ht = d.htp                         # (per-direction size AS COMPUTED: every offset / kernel size below is in the padded layout)
        Tn, MT = self.Tn, self.MT
        dt = e.dt
        drop = training and d.dropout > 0
        MAIN, AUX = 0, 2
        e._sid = AUX
        e._wait(P, "fwd_begin")
        # the shared target embedding table is updated by the side half of Adam: its own event when that half runs it first
        # (cond_emb_fg, the default: the table is updated by the FOREGROUND half of the optimiser step, i.e. before this plan starts; updating
        #  it first in the background half and waiting for that alone was worth 3.524 -> 3.507 ms, this 3.518 -> 3.483)
        if not e.cond_emb_fg:
            e._wait(P, "side_fwd")
        # (the lazily updated table: this batch's rows were brought up to date on THIS stream, in front of the decoder's lookup: Workspace.side_dec_gx)
        e._call(P, lib.vmmt_gather_rows, dt, e.pp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                self.tgt_bt.data_ptr(), self.Yt.p(), self.Yt.ld, MT, E)
        x, xcols = self.Yt, E
        for l in range(Lyr):
            wih, bsum = e.sh["enct_wih_l%d" % l], e.sh["enct_b_l%d" % l]
            e._gemm(P, L.GEMM_NT, x.p(), x.ld, wih.p(), wih.ld, self.enct_gx[l].p(), self.enct_gx[l].ld, MT, 8 * ht, xcols,
                    addend=bsum.p(), ld_add=bsum.ld, add_rows=1, out_f32=1)
            arr = (L.LstmDirFwd * (2 * B))()                    # the whole recurrence is ONE host call (vmmt_lstm_chain_fwd)
            for step in range(B):
                for k in range(2):
                    t = step if k == 0 else B - 1 - step           # "time" = sentence index (H5)
                    tp = (t - 1) if k == 0 else (t + 1)
                    first = step == 0
                    whh = e.sh["enct_whh_l%d_d%d" % (l, k)]
                    a = arr[2 * step + k]
                    if first:
                        a.h_prev, a.ld_hprev = self.enct_hzero.p(0, k * ht), self.enct_hzero.ld
                    else:
                        a.h_prev, a.ld_hprev = self.enct_out[l].p(tp * Tn, k * ht), self.enct_out[l].ld
                        a.c_prev = self.enct_c[l].p(tp * Tn, k * ht)
                    a.ld_cprev = self.enct_c[l].ld
                    a.w_hh, a.ld_w = whh.p(), whh.ld
                    a.gx, a.ld_gx = self.enct_gx[l].p(t * Tn, k * 4 * ht), self.enct_gx[l].ld
                    a.gates, a.ld_gates = self.enct_gates[l].p(t * Tn, k * 4 * ht), self.enct_gates[l].ld
                    a.c_out, a.ld_c = self.enct_c[l].p(t * Tn, k * ht), self.enct_c[l].ld
                    a.h_out, a.ld_h = self.enct_out[l].p(t * Tn, k * ht), self.enct_out[l].ld
                    a.t, a.capture = t, 0
            e._lstm_seq_fwd(P, arr, 2, B, None, Tn, ht)
            x, xcols = self.enct_out[l], 2 * ht
            if l < Lyr - 1 and drop:
                e._call(P, lib.vmmt_mul, dt, self.enct_out[l].p(), self.enct_out[l].ld, self.enct_mask[l].p(), self.enct_mask[l].ld,
                        self.enct_xdrop[l].p(), self.enct_xdrop[l].ld, MT, 2 * ht)
                x = self.enct_xdrop[l]
        e._record(P, "enct_done")

    def _cond_forward(self, P, training, ctx):
        """main stream: h_x, h_y, v -> p(z|x) and q(z|x,y,v)  (Models.py:883-914)"""
        e, d, lib = self.e, self.e.d, self.e.lib
        B, S, H, E, Z, D, Lyr = self.B, self.S, d.hid, d.emb, d.z, d.img, d.layers
        Hp, Ht = d.hp, 2 * d.htp            # column ranges of the q-network input as computed: [h_x : Hp | h_y : Ht | v : D]
        Tn, MT = self.Tn, self.MT
        dt = e.dt
        MAIN = 0
        e._sid = MAIN
        # [h_x ; h_y ; v]: the two means are written straight into their column ranges of the q-network input
        e._call(P, lib.vmmt_masked_mean, dt, ctx.p(), ctx.ld, self.src_len.data_ptr(), self.hq.p(0, 0), self.hq.ld, B, S, H)
        e._call(P, lib.vmmt_pack, dt, self.img.p(), None, self.img.ld, self.hq.p(0, Hp + Ht), self.hq.ld, B, D, 0)
        # p(z|x) = gen_net_global(h_x) (the values of h_x are those of the detached copy)
        for br, outb, act in (("location", self.mu_p, L.ACT_NONE), ("scale", self.sigma_p, L.ACT_SOFTPLUS)):
            w1, w2 = e.sh["p_%s_w1" % br], e.sh["p_%s_w2" % br]
            e._gemm(P, L.GEMM_NT, self.hq.p(), self.hq.ld, w1.p(), w1.ld, self.p_h1[br].p(), self.p_h1[br].ld, B, Z, H,
                    addend=e.pp("gen_net_global.%s.fc1.bias" % br), ld_add=Z, add_rows=1, act=L.ACT_RELU)
            e._gemm(P, L.GEMM_NT, self.p_h1[br].p(), self.p_h1[br].ld, w2.p(), w2.ld, outb.p(), outb.ld, B, Z, Z,
                    addend=e.pp("gen_net_global.%s.fc2.bias" % br), ld_add=Z, add_rows=1, act=act, out_f32=1)
        e._wait(P, "enct_done")
        e._call(P, lib.vmmt_masked_mean_bm, dt, self.enct_out[Lyr - 1].p(), self.enct_out[Lyr - 1].ld, self.tgt_len.data_ptr(),
                self.hq.p(0, Hp), self.hq.ld, B, Tn, Ht)
        # q(z|x,y,v) sits on the step's critical path (behind encoder_tgt's recurrence) and its two branches are 16-workgroup products over
        # K = 2H + D: the scale branch runs on the side stream next to the location branch (as in the fixed-prior model's unfused path)
        par = bool(e.q_parallel and e.use_side_stream)
        if par:
            e._record(P, "hq_ready")
        for br, outb, act in (("scale", self.sigma, L.ACT_SOFTPLUS), ("location", self.mu, L.ACT_NONE)):
            w1, w2 = e.sh["q_%s_w1" % br], e.sh["q_%s_w2" % br]
            if par and br == "scale":
                e._sid = 1
                e._wait(P, "hq_ready")
            e._gemm(P, L.GEMM_NT, self.hq.p(), self.hq.ld, w1.p(), w1.ld, self.q_h1[br].p(), self.q_h1[br].ld, B, Z, d.qin_p,
                    addend=e.pp("inf_net_global.%s.fc1.bias" % br), ld_add=Z, add_rows=1, act=L.ACT_RELU)
            e._gemm(P, L.GEMM_NT, self.q_h1[br].p(), self.q_h1[br].ld, w2.p(), w2.ld, outb.p(), outb.ld, B, Z, Z,
                    addend=e.pp("inf_net_global.%s.fc2.bias" % br), ld_add=Z, add_rows=1, act=act, out_f32=1)
            if par and br == "scale":
                e._record(P, "sigma_ready")
                e._sid = MAIN
        if par:
            e._wait(P, "sigma_ready")

    def _cond_backward(self, P, drop):
        """aux stream, right behind vmmt_latent_cond_bwd: backward of p(z|x); d h_x goes to the main stream (event dhbar_p)"""
        e, d, lib = self.e, self.e.d, self.e.lib
        B, H, Z = self.B, d.hid, d.z
        dt = e.dt
        for i, (br, dy) in enumerate((("location", self.p_dmu), ("scale", self.p_dpre))):
            w1, w2 = e.sh["p_%s_w1" % br], e.sh["p_%s_w2" % br]
            pre = "gen_net_global.%s" % br
            e._gemm(P, L.GEMM_TN, dy.p(), dy.ld, self.p_h1[br].p(), self.p_h1[br].ld, e.gp(pre + ".fc2.weight"), Z, Z, Z, B, out_f32=1, split_k=-1)
            e._call(P, lib.vmmt_colsum, dt, dy.p(), dy.ld, B, Z, 0, 0, e.gp(pre + ".fc2.bias"), None)
            e._gemm(P, L.GEMM_NN, dy.p(), dy.ld, w2.p(), w2.ld, self.p_dh1[br].p(), self.p_dh1[br].ld, B, Z, Z)
            e._call(P, lib.vmmt_act_bwd, dt, L.ACT_RELU, self.p_dh1[br].p(), self.p_dh1[br].ld, 0, self.p_h1[br].p(), self.p_h1[br].ld,
                    None, 0, self.p_dh1[br].p(), self.p_dh1[br].ld, B, Z)
            e._gemm(P, L.GEMM_TN, self.p_dh1[br].p(), self.p_dh1[br].ld, self.hq.p(), self.hq.ld, e.gp(pre + ".fc1.weight"), H,
                    Z, H, B, out_f32=1, split_k=-1)
            e._call(P, lib.vmmt_colsum, dt, self.p_dh1[br].p(), self.p_dh1[br].ld, B, Z, 0, 0, e.gp(pre + ".fc1.bias"), None)
            e._gemm(P, L.GEMM_NN, self.p_dh1[br].p(), self.p_dh1[br].ld, w1.p(), w1.ld, self.dhbar_p.p(), self.dhbar_p.ld, B, H, Z,
                    accumulate=1 if i else 0)
        e._record(P, "dhbar_p")

    def _cond_backward_tgt(self, P, drop):
        """stream TGT: d h_y -> encoder_tgt (BPTT over the B recurrent steps) -> its parameters and the shared target embeddings.
        (Cutting the recurrence into 2 / 4 / 8 launches -- vmmt_lstm_seq_bwd continues a chain -- with the parameter gradients of a
        finished piece issued next to the rest was measured: 4.36 / 4.38 / 4.59 ms per step against 4.27 ms in one piece; every
        relaunch costs more than the shorter tail saves.)"""
        e, d, lib = self.e, self.e.d, self.e.lib
        B, H, E, Lyr = self.B, d.hid, d.emb, d.layers
        ht, ht_t = d.htp, d.ht              # per-direction size as computed / as stored in the arena
        tmap = (ht, ht_t)                   # padded gate blocks -> nn.LSTM's rows; padded direction blocks -> the layer input's columns
        Tn, MT = self.Tn, self.MT
        dt = e.dt
        e._call(P, lib.vmmt_masked_mean_bwd, dt, self.dhy.p(), self.dhy.ld, self.tgt_len.data_ptr(), self.enct_dout.p(),
                self.enct_dout.ld, B, Tn, 2 * ht, 1, 0)
        dh_above = self.enct_dout
        for l in reversed(range(Lyr)):
            dg = self.enct_dgates[l]
            e._zero(P, [self.enct_dcc[l].t])
            arr = (L.LstmDirBwd * (2 * B))()
            for step in range(B):
                for k in range(2):
                    t = (B - 1 - step) if k == 0 else step
                    tn = (t + 1) if k == 0 else (t - 1)
                    tp = (t - 1) if k == 0 else (t + 1)
                    a = arr[2 * step + k]
                    whhT = e.sh["enct_whhT_l%d_d%d" % (l, k)]
                    if step > 0:
                        a.dgates_next, a.ld_dgn = dg.p(tn * Tn, k * 4 * ht), dg.ld
                    a.w_hh_t, a.ld_wt = whhT.p(), whhT.ld
                    a.dh_above, a.ld_dha = dh_above.p(t * Tn, k * ht), dh_above.ld
                    a.gates, a.ld_gates = self.enct_gates[l].p(t * Tn, k * 4 * ht), self.enct_gates[l].ld
                    a.c_t, a.ld_ct = self.enct_c[l].p(t * Tn, k * ht), self.enct_c[l].ld
                    if 0 <= tp < B:
                        a.c_prev, a.ld_cp = self.enct_c[l].p(tp * Tn, k * ht), self.enct_c[l].ld
                    a.dc_carry, a.ld_dcc = self.enct_dcc[l].p(0, k * ht), self.enct_dcc[l].ld
                    a.dgates_out, a.ld_dgo = dg.p(t * Tn, k * 4 * ht), dg.ld
                    a.t, a.inject = t, 0
            e._lstm_seq_bwd(P, arr, 2, B, None, Tn, ht)
            wih = e.sh["enct_wih_l%d" % l]
            if l > 0:
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, wih.p(), wih.ld, self.enct_dx[l - 1].p(), self.enct_dx[l - 1].ld, MT, 2 * ht, 8 * ht)
                if drop:
                    e._call(P, lib.vmmt_mul, dt, self.enct_dx[l - 1].p(), self.enct_dx[l - 1].ld, self.enct_mask[l - 1].p(),
                            self.enct_mask[l - 1].ld, self.enct_dx[l - 1].p(), self.enct_dx[l - 1].ld, MT, 2 * ht)
                dh_above = self.enct_dx[l - 1]
            xin = (self.Yt if l == 0 else (self.enct_xdrop[l - 1] if drop else self.enct_out[l - 1]))
            xcols_t = E if l == 0 else H                               # columns of weight_ih as stored ...
            xcols, xmap = (E, None) if l == 0 else (2 * ht, tmap)      # ... and of the layer input as computed
            for k, suf in enumerate(("", "_reverse")):
                gw = "encoder_tgt.rnn.weight_hh_l%d%s" % (l, suf)
                if B > 1:
                    if k == 0:   # h_prev[t] = out[t-1]
                        e._gemm(P, L.GEMM_TN, dg.p(Tn, k * 4 * ht), dg.ld, self.enct_out[l].p(0, k * ht), self.enct_out[l].ld, e.gp(gw), ht_t,
                                4 * ht, ht_t, (B - 1) * Tn, out_f32=1, split_k=-1, rmap=tmap)
                    else:        # h_prev[t] = out[t+1]
                        e._gemm(P, L.GEMM_TN, dg.p(0, k * 4 * ht), dg.ld, self.enct_out[l].p(Tn, k * ht), self.enct_out[l].ld, e.gp(gw), ht_t,
                                4 * ht, ht_t, (B - 1) * Tn, out_f32=1, split_k=-1, rmap=tmap)
                e._call(P, lib.vmmt_colsum, dt, dg.p(0, k * 4 * ht), dg.ld, MT, 4 * ht, ht if ht != ht_t else 0, ht_t,
                        e.gp("encoder_tgt.rnn.bias_ih_l%d%s" % (l, suf)), e.gp("encoder_tgt.rnn.bias_hh_l%d%s" % (l, suf)))
                e._gemm(P, L.GEMM_TN, dg.p(0, k * 4 * ht), dg.ld, xin.p(), xin.ld, e.gp("encoder_tgt.rnn.weight_ih_l%d%s" % (l, suf)), xcols_t,
                        4 * ht, xcols, MT, out_f32=1, split_k=-1, rmap=tmap, cmap=xmap)
            if l == 0:   # shared table (ModelConstructor.py:456-457): scatter-add next to the decoder's contribution; pad row skipped
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, wih.p(), wih.ld, e.gp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                        MT, E, 8 * ht, out_f32=1, scatter_ids=self.tgt_bt.data_ptr())
