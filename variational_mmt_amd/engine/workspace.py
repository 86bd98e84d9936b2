"""Per-shape device buffers and the forward / loss launch plans (the backward plan: backward.py; the `--conditional` pieces:
conditional.py)."""
import ctypes as C
import math
from os import environ as _os_env

import torch

from .. import _lib as L
from .backward import BackwardPlan
from .conditional import ConditionalPlans
from .layout import Buf, KPAD, PAD, _ru  # noqa: F401


class Workspace(BackwardPlan, ConditionalPlans):
    """All per-shape device buffers + the launch plans for (B sentences, S source positions, T' decoder steps)."""

    def __init__(self, eng, B, S, Tp):
        self.e, self.B, self.S, self.Tp = eng, B, S, Tp
        d, T, dev = eng.d, eng.T, eng.dev
        f32, i64 = torch.float32, torch.int64
        H, Hd, E, Z, D, V, Lyr, dirs = d.hid, d.hd, d.emb, d.z, d.img, d.vt, d.layers, d.dirs
        Hp, Hdp = d.hp, d.hdp         # hidden sizes as computed (Dims.hp): gate g of a 4H vector at column g * Hp; [c ; r] halves at 0 / Hp
        M, MS = Tp * B, S * B
        self.M, self.MS = M, MS
        nb = lambda r, c, dt=T, **kw: Buf(r, c, dt, dev, **kw)
        # inputs
        self.src = torch.zeros(MS, dtype=i64, device=dev)
        self.tgt_in = torch.zeros(M, dtype=i64, device=dev)
        self.y = torch.zeros(M, dtype=i64, device=dev)
        self.src_len = torch.zeros(B, dtype=i64, device=dev)
        self.img_idx = torch.zeros(B, dtype=i64, device=dev)
        self.img = nb(B, D, f32)
        self.eps = nb(B, Z, f32, ld=Z)
        self.stats = torch.zeros(L.STAT_COUNT, dtype=f32, device=dev)
        self.kl_global = torch.zeros(1, dtype=f32, device=dev)
        # encoder
        self.Xs = nb(MS, E)
        self.enc_gx = [nb(MS, dirs * 4 * Hdp, f32) for _ in range(Lyr)]
        self.enc_gates = [nb(MS, dirs * 4 * Hdp) for _ in range(Lyr)]
        self.enc_c = [nb(MS, H, f32) for _ in range(Lyr)]
        self.enc_out = [nb(MS, H) for _ in range(Lyr)]
        self.enc_mask = [nb(MS, H) if (d.dropout > 0 and l < Lyr - 1) else None for l in range(Lyr)]
        self.enc_xdrop = [nb(MS, H) if (d.dropout > 0 and l < Lyr - 1) else None for l in range(Lyr)]
        self.hn = [nb(B, H) for _ in range(Lyr)]
        self.cn = [nb(B, H, f32) for _ in range(Lyr)]
        # q(z|x)
        self.hbar = nb(B, H)
        self.q_h1 = {br: nb(B, d.zp) for br in ("location", "scale")}     # (the fused q(z|x) kernel stores whole 128-column tiles: zeros beyond Z)
        self.mu = nb(B, Z, f32, ld=Z)
        self.sigma = nb(B, Z, f32, ld=Z)
        self.z32 = nb(B, Z, f32, ld=Z)
        self.zT = nb(B, Z)
        self.kl_b = torch.zeros(B, dtype=f32, device=dev)
        # decoder
        # target embeddings [M][E], and -- in the same rows, from column Ep on -- the sample z repeated over the decoder steps: the
        # backward's dW_ih = dgates^T [emb ; z] (VI_Model1.py:99-100: the decoder input is the concatenation) is then ONE product over
        # the rows of this buffer instead of a second one that walks z with a row modulus (2.6 MB at the benchmark shape; the FORWARD
        # still adds z W_z once per sentence and never reads these columns)
        self.Ep = _ru(E, KPAD)
        self.z_in_Xt = bool(Z <= E)
        self.Xt = nb(M, E, ld=self.Ep + _ru(Z, KPAD)) if self.z_in_Xt else nb(M, E)
        self.zrep_ids = (torch.arange(M, dtype=i64) % B).to(dev) if self.z_in_Xt else None
        self.zx = nb(B, 4 * Hp, f32)
        self.dec_gx = [nb(M, 4 * Hp, f32) for _ in range(Lyr)]
        self.dec_gates = [nb(M, 4 * Hp) for _ in range(Lyr)]
        self.dec_c = [nb(M, H, f32) for _ in range(Lyr)]
        self.dec_out = [nb(M, H) for _ in range(Lyr - 1)]
        self.dec_mask = [nb(M, H) if d.dropout > 0 else None for _ in range(Lyr - 1)]
        self.dec_xdrop = [nb(M, H) if d.dropout > 0 else None for _ in range(Lyr - 1)]
        self.cat = nb(M, 2 * Hp)
        self.Q = nb(M, H)
        self.probs = torch.zeros(M * S, dtype=f32, device=dev)
        self.AH = nb(M, H)
        self.out_mask = nb(M, H) if d.dropout > 0 else None
        self.O = nb(M, H) if d.dropout > 0 else self.AH
        # image network
        self.gate = torch.zeros(B, dtype=f32, device=dev)
        self.zt = nb(B, Z)
        self.h1v = nb(B, D)
        self.mu_v = nb(B, D, f32)
        # loss
        self.npart = eng.lib.vmmt_gen_npart(V)
        self.part_max = torch.zeros(self.npart * M, dtype=f32, device=dev)
        self.part_sum = torch.zeros(self.npart * M, dtype=f32, device=dev)
        self.tgt_logit = torch.zeros(M, dtype=f32, device=dev)
        self.lse = torch.zeros(M, dtype=f32, device=dev)
        self.tok_nll = torch.zeros(M, dtype=f32, device=dev)
        # backward
        # G^T [V][T'B] is the largest buffer of a step (307 MB at B 256 / V 30 000 / T' 20): ONE allocation shared by all
        # workspaces.  Leftovers of another shape are harmless: it is fully rewritten for columns < M by the generator backward
        # before anything reads it, and its padding only ever meets the zero padding of O (dW_g, K = M) or of W_g (dO, K = V).
        # Fused generator passes (csrc/generator_fused.hip) where they apply: the statistics pass also produces dO, a second pass
        # dWg / db, and G^T is never formed.  Otherwise the G^T path of csrc/generator.hip.
        self.dO32 = nb(M, H, f32)
        self.dXs, self.dXt = nb(MS, E, f32), nb(M, E, f32)          # d(embedding rows) before their scatter into the tables' gradients
        wg_ld = eng.sh["wg"].ld
        self.gen_fused = bool(eng.gen_fused and eng.lib.vmmt_gen_fused_applies(eng.dt, wg_ld, self.O.ld, M, V, _ru(H, KPAD)))
        if self.gen_fused:
            nws = int(eng.lib.vmmt_gen_fused_ws_floats(M, V, _ru(H, KPAD)))
            self.gen_ws = eng.shared_storage("gen_ws", nws, f32)          # vocabulary-slice partials: shared by all workspaces
            self.gen_y32 = torch.zeros(_ru(M, 32), dtype=torch.int32, device=dev)
            self.GT = None
            ns, vps, mpad = C.c_int(), C.c_int(), C.c_int64()
            L.check(eng.lib.vmmt_gen_fused_geometry(M, V, _ru(H, KPAD), C.byref(ns), C.byref(vps), C.byref(mpad)), "vmmt_gen_fused_geometry")
            self.gen_ns, self.gen_vps, self.gen_mpad = ns.value, vps.value, mpad.value
            Mk = _ru(M, KPAD) + KPAD
            self.gen_ldp = _ru(V, 32)
            # P [M][V] bf16 (the forward sweep's softmax weights: 307 MB at B 256 / V 30 000 / T' 20): one allocation shared by all
            # workspaces; leftovers of other shapes are finite and only ever meet the zero rows of O'_s
            self.gen_P = eng.shared_storage("gen_P", Mk * self.gen_ldp, T)
            self.gen_cs = torch.zeros(self.gen_ns * self.gen_mpad, dtype=f32, device=dev)
            self.gen_Os = torch.zeros(self.gen_ns, Mk, _ru(H, KPAD), dtype=T, device=dev)
            # COMPACTED tokens (Engine.gen_compact; vmmt_compact_nonpad): when the forward is told how many decoder rows carry a target, the
            # generator's calls run over that many tokens (rounded up to 128) instead of all T' B rows.  The vocabulary slices follow the
            # token count (fewer token blocks -> more slices), so the per-slice buffers are sized for the worst count
            self.gen_rows = torch.full((M + 256,), -1, dtype=torch.int32, device=dev)
            self.gen_cnt = torch.zeros(2, dtype=torch.int32, device=dev)      # [token rows of the last step, sticky "rows were left out" flag]
            self.gen_geo = {}
            Kp = _ru(H, KPAD)
            need_cs, need_os, need_ws = self.gen_ns * self.gen_mpad, self.gen_ns * Mk * Kp, nws
            if eng.gen_compact:
                for Mc in range(1024, M, 128):
                    L.check(eng.lib.vmmt_gen_fused_geometry(Mc, V, Kp, C.byref(ns), C.byref(vps), C.byref(mpad)), "vmmt_gen_fused_geometry")
                    self.gen_geo[Mc] = (ns.value, vps.value, mpad.value)
                    need_cs = max(need_cs, ns.value * mpad.value)
                    need_os = max(need_os, ns.value * (Mc + KPAD) * Kp)
                    need_ws = max(need_ws, int(eng.lib.vmmt_gen_fused_ws_floats(Mc, V, Kp)))
                if need_ws > nws:
                    self.gen_ws = eng.shared_storage("gen_ws", need_ws, f32)
                if need_cs > self.gen_cs.numel():
                    self.gen_cs = torch.zeros(need_cs, dtype=f32, device=dev)
                if need_os > self.gen_Os.numel():
                    self.gen_Os_flat = torch.zeros(need_os, dtype=T, device=dev)
                    self.gen_Os = self.gen_Os_flat[:self.gen_ns * Mk * Kp].view(self.gen_ns, Mk, Kp)
            self.gen_Mc = M                   # token count of the current step's generator calls (M: not compacted)
        else:
            self.GT = Buf(V, M, T, dev, storage=eng.shared_storage("GT", Buf.elems(V, M), T))
        self.gen_Mc = getattr(self, "gen_Mc", M)
        self.dPre = nb(M, H)
        self.dcat = nb(M, 2 * Hp)
        self.dQ = nb(M, H)
        self.dctx = nb(MS, H)
        self.dR = nb(M, H)
        self.dec_dgates = [nb(M, 4 * Hp) for _ in range(Lyr)]
        self.dec_dcc = [nb(B, H, f32) for _ in range(Lyr)]
        self.dec_dh0 = [nb(B, H, f32) for _ in range(Lyr)]
        self.dec_dx = [nb(M, H) for _ in range(Lyr - 1)]
        self.enc_dgates = [nb(MS, dirs * 4 * Hdp) for _ in range(Lyr)]
        self.enc_dcc = [nb(B, H, f32) for _ in range(Lyr)]
        self.enc_dx = [nb(MS, H) for _ in range(Lyr - 1)]
        self.q_dmu = nb(B, Z)
        self.q_dpre = nb(B, Z)
        self.q_dh1 = {br: nb(B, Z) for br in ("location", "scale")}
        self.dmu_v = nb(B, D)
        self.dh1v = nb(B, D)
        self.dh1v32 = nb(B, D, f32)
        self.dzt = nb(B, Z, f32)
        if d.conditional:
            self._cond_alloc()
        self._keep = []
        self.events = {}
        self.plan_fwd_train = self._plan_forward(True)
        self.plan_fwd_eval = self._plan_forward(False)
        self.plan_loss_train = self._plan_loss(True)
        self.plan_loss_eval = self._plan_loss(False)
        self.plan_bwd = None
        self._bwd_key = None

    # ---------------------------------------------------------------------------------------------- forward plan
    def _plan_forward(self, training):
        e, d, lib = self.e, self.e.d, self.e.lib
        B, S, Tp, M, MS = self.B, self.S, self.Tp, self.M, self.MS
        H, Hd, E, Z, D, V, Lyr, dirs = d.hid, d.hd, d.emb, d.z, d.img, d.vt, d.layers, d.dirs
        Hp, Hdp = d.hp, d.hdp
        dt = e.dt
        P = []
        drop = training and d.dropout > 0
        MAIN, SIDE = 0, 1
        e._sid = MAIN
        e._record(P, "fwd_begin")
        # (bg_after_head) the head of the main stream first -- source rows and the encoder's first input projection -- and only behind it the
        # side-stream half of the last optimiser step (BG_FLUSH) and the side stream's own work, the decoder's input projection first
        early_head = bool(training and e.bg_after_head and not d.conditional and e.use_side_stream)
        # the source embeddings: the A operand of the encoder's first input projection is fetched from the table's bf16 copy by token id INSIDE
        # the product (no lookup launch, no [S B x E] copy in front of it); the copy the backward multiplies (dW_ih) is gathered later, on
        # the side stream.  Training plans of the lazily updated tables only: their catch-up keeps the copy current for the batch's rows
        emb_tab = e.row_shadow(0) if training else None

        def src_gather():
            e._call(P, lib.vmmt_gather_rows, dt, e.pp("encoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                    self.src.data_ptr(), self.Xs.p(), self.Xs.ld, MS, E)

        def enc_gx0():
            wih, bsum = e.sh["enc_wih_l0"], e.sh["enc_b_l0"]
            if emb_tab is not None:
                e._gemm(P, L.GEMM_NT, emb_tab.p(), emb_tab.ld, wih.p(), wih.ld, self.enc_gx[0].p(), self.enc_gx[0].ld, MS, dirs * 4 * Hdp,
                        E, addend=bsum.p(), ld_add=bsum.ld, add_rows=1, out_f32=1, a_row_ids=self.src.data_ptr())
            else:
                e._gemm(P, L.GEMM_NT, self.Xs.p(), self.Xs.ld, wih.p(), wih.ld, self.enc_gx[0].p(), self.enc_gx[0].ld, MS, dirs * 4 * Hdp,
                        E, addend=bsum.p(), ld_add=bsum.ld, add_rows=1, out_f32=1)
        if early_head:
            e._row_mark_entries(P, 0, self.src.data_ptr(), MS, with_shadow=emb_tab is not None)
            if emb_tab is None:
                src_gather()
            enc_gx0()
            e._record(P, "fwd_begin")
            P.append((None, None, "BG_FLUSH", None, MAIN))

        def side_dec_gx():
            if training:
                # the lazily updated embedding tables (Engine._build_row_tables): flag this batch's rows, bring them up to date and clear
                # their gradient rows IN FRONT of the lookup; this stream carried the table's share of the last update
                if d.conditional and e.rows_active():
                    # encoder_tgt reads the same table by the transposed target's ids (every position, pad fill included): flagged by a
                    # launch of their own in front of the table's ONE catch-up -- on the AUX stream, at the head of encoder_tgt's chain (the
                    # conditional step's longest: behind this stream's share of the last update it started 0.1 ms later); this stream's
                    # lookup waits for `tgt_rows`
                    sid_ = e._sid
                    e._sid = 2
                    e._wait(P, "fwd_begin")
                    if not e.cond_emb_fg:
                        e._wait(P, "side_fwd")
                    t_ = e.row_tables[1]
                    e._call(P, lib.vmmt_rows_mark, self.tgt_bt.data_ptr(), self.MT, t_["flags"].data_ptr(), t_["R"], t_["hist"].data_ptr())
                    e._row_mark_entries(P, 1, self.tgt_in.data_ptr(), M)
                    e._record(P, "tgt_rows")
                    e._sid = sid_
                    e._wait(P, "tgt_rows")
                else:
                    e._row_mark_entries(P, 1, self.tgt_in.data_ptr(), M)
            e._call(P, lib.vmmt_gather_rows, dt, e.pp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                    self.tgt_in.data_ptr(), self.Xt.p(), self.Xt.ld, M, E)
            we = e.sh["dec_wih_l0_e"]
            e._gemm(P, L.GEMM_NT, self.Xt.p(), self.Xt.ld, we.p(), we.ld, self.dec_gx[0].p(), self.dec_gx[0].ld, M, 4 * Hp, E, out_f32=1)
            e._record(P, "dec_gx")
        # ---- side stream, underneath the encoder: zero the gradient arena (every gradient writer of the backward plan
        #      runs on the side stream), target embeddings and the time-parallel part of the decoder input projection
        e._sid = SIDE
        e._wait(P, "fwd_begin")
        e._record(P, "side_fwd")
        if early_head:
            # the generator's third of the optimiser step, the gradient zeroing behind it and the small preparations of the loss move to
            # the AUX stream (idle in the forward): the side stream keeps the decoder's input projection and the image network
            side_dec_gx()
            e._sid = 2
            e._wait(P, "dec_gx")
            P.append((None, None, "BG_FLUSH2", None, 2))
        elif e.dec_gx_first:
            side_dec_gx()
        # the side stream's chain -- the last update's background half, the decoder's input projection, zeroing, masks, the image network -- ends
        # where the main stream's does (`img_fwd`, in front of the sweep) and is the longer of the two (tools/critical.py: the step without the
        # zeroing is 27 us shorter, without the encoder's recurrence 3 us): the zeroing and the masks go to the AUX stream, idle at that point
        # (not under data parallelism: one-rank RCCL rehearsal 2.07 against 1.93 ms, and 2.02 against 1.89 once the KL all-reduce had gone from
        #  the head of the aux stream's chain;
        #  not for the conditional model, whose aux stream starts the step's longest chain, encoder_tgt's backward: 2.77 against 2.68 ms)
        aux_zero = bool(not early_head and e.dec_gx_first and e.zero_on_aux and e.use_side_stream and e.use_aux_stream and training and not e.dp_on() and
                        not d.conditional)
        if aux_zero:
            e._sid = 2
            e._wait(P, "side_fwd")       # (behind the side-stream half of the last update, which reads the gradients)
        if training:
            # the generator weight gradient (first in the arena, a third of it) is WRITTEN by its one GEMM, not accumulated
            # ... together with the small accumulators of the backward plan (off the critical path instead of in front of
            # their users): one launch
            if e.rows_active():        # the tables' gradient rows are cleared row by row (vmmt_rows_catchup)
                g_ranges, lo = [], e.offsets["generator.0.bias"][0]
                for t in sorted(e.row_tables, key=lambda t: t["off"]):
                    g_ranges.append(e.flat_g[lo:t["off"]])
                    lo = t["end"]
                g_ranges.append(e.flat_g[lo:])
                g_ranges = [r for r in g_ranges if r.numel()]
            else:
                g_ranges = [e.flat_g[e.offsets["generator.0.bias"][0]:]]
            compact = bool(self.gen_fused and e.gen_compact)       # (rows of dO / lse / tok_nll without a token are not written then)
            e._zero(P, g_ranges + [e._sumsq[:L.SUMSQ_SLOTS], self.dh1v32.t, self.dzt.t] +
                    ([self.dO32.t, self.lse, self.tok_nll] if compact else [] if self.gen_fused else [self.dO32.t]) +       # (fused generator: dO is stored, not accumulated)
                    [b.t for l in range(Lyr) for b in (self.dec_dcc[l], self.enc_dcc[l])])
            # `grad_zero`: what accumulates into the gradient arena from another stream than this one waits for it (the backward plan's KL /
            # q(z|x) chain on the aux stream; everything else of the backward is behind the forward plan's last join)
            e._record(P, "grad_zero")
            self._zero_sid = e._sid
            if compact:
                self._compact_entry = len(P)
                e._call(P, lib.vmmt_compact_nonpad, self.y.data_ptr(), M, PAD, M, self.gen_rows.data_ptr(), self.gen_cnt.data_ptr())
        self._mask_entries = getattr(self, "_mask_entries", {})
        if drop:
            # output dropout mask (VI_Model1.py:132): only needed after the decoder -> generated in the background
            self._mask_entries["dec_out"] = (len(P), self.out_mask)
            e._call(P, lib.vmmt_dropout_mask, dt, self.out_mask.p(), self.out_mask.rows * self.out_mask.ld, d.dropout, 0)
            e._record(P, "out_mask")
        if early_head or aux_zero:
            e._record(P, "aux_fwd")
            e._sid = SIDE
        elif not e.dec_gx_first:
            side_dec_gx()
        if d.conditional:
            self._cond_forward_aux(P, training)
        # a1 image rows (TrainerMultimodal.py:632-639) -- table pointer is patched in at run time (set_image_table).  Fixed prior: only
        # the loss reads them, so the gather runs on the side stream (joined by "img_fwd" at the end of the plan); the conditional
        # model feeds them to q(z|x,y,v) and keeps it at the head of the main stream
        if not hasattr(self, '_img_idx'):
            self._img_idx = {}
        if d.conditional:
            e._sid = MAIN
        self._img_idx[bool(training)] = len(P)
        e._call(P, lib.vmmt_gather_rows, L.F32, None, D, self.img_idx.data_ptr(), self.img.p(), self.img.ld, B, D)
        e._sid = MAIN
        # a2 source embeddings
        if not early_head:
            if training:
                e._row_mark_entries(P, 0, self.src.data_ptr(), MS, with_shadow=emb_tab is not None)
            if emb_tab is None:
                src_gather()
        # a3 encoder
        x, xcols = self.Xs, E
        for l in range(Lyr):
            wih, bsum = e.sh["enc_wih_l%d" % l], e.sh["enc_b_l%d" % l]
            if l == 0:
                if not early_head:
                    enc_gx0()
            else:
                e._gemm(P, L.GEMM_NT, x.p(), x.ld, wih.p(), wih.ld, self.enc_gx[l].p(), self.enc_gx[l].ld, MS, dirs * 4 * Hdp,
                        xcols, addend=bsum.p(), ld_add=bsum.ld, add_rows=1, out_f32=1)
            seq = (L.LstmDirFwd * (S * dirs))()                 # the whole recurrence: step-major, then direction
            if not hasattr(self, "hzero"):
                self.hzero = Buf(B, H, e.T, e.dev)              # first step: h_prev reads zeros
            for step in range(S):
                for k in range(dirs):
                    t = step if k == 0 else S - 1 - step
                    tp = (t - 1) if k == 0 else (t + 1)
                    first = step == 0
                    whh = e.sh["enc_whh_l%d_d%d" % (l, k)]
                    a = seq[step * dirs + k]
                    a.h_prev = self.enc_out[l].p(tp * B, k * Hdp) if not first else self.enc_out[l].p(t * B, k * Hdp)
                    a.ld_hprev = self.enc_out[l].ld
                    a.c_prev = None if first else self.enc_c[l].p(tp * B, k * Hdp)
                    a.ld_cprev = self.enc_c[l].ld
                    a.w_hh, a.ld_w = whh.p(), whh.ld
                    a.gx, a.ld_gx = self.enc_gx[l].p(t * B, k * 4 * Hdp), self.enc_gx[l].ld
                    a.gates, a.ld_gates = self.enc_gates[l].p(t * B, k * 4 * Hdp), self.enc_gates[l].ld
                    a.c_out, a.ld_c = self.enc_c[l].p(t * B, k * Hdp), self.enc_c[l].ld
                    a.h_out, a.ld_h = self.enc_out[l].p(t * B, k * Hdp), self.enc_out[l].ld
                    a.h_n, a.ld_hn = self.hn[l].p(0, k * Hdp), self.hn[l].ld
                    a.c_n, a.ld_cn = self.cn[l].p(0, k * Hdp), self.cn[l].ld
                    a.t = t
                    a.capture = 1 if k == 0 else 2
                    if first:
                        a.h_prev, a.ld_hprev = self.hzero.p(0, k * Hdp), self.hzero.ld
            e._lstm_seq_fwd(P, seq, dirs, S, self.src_len.data_ptr(), B, Hdp)
            x, xcols = self.enc_out[l], H
            if l < Lyr - 1 and drop:
                e._call(P, lib.vmmt_mul, dt, self.enc_out[l].p(), self.enc_out[l].ld, self.enc_mask[l].p(), self.enc_mask[l].ld,
                        self.enc_xdrop[l].p(), self.enc_xdrop[l].ld, MS, H)
                x = self.enc_xdrop[l]
        ctx = self.enc_out[Lyr - 1]
        if d.conditional:
            self._cond_forward(P, training, ctx)
        # a4 + a5 fused: masked mean -> both MLPs -> sample -> KL in ONE launch (csrc/qnet.hip); the separate kernels below remain for
        # fp32 parity mode, the conditional model and sizes the fused kernel does not take
        Zp = d.zp
        fused_q = (e.fused_qnet and not d.conditional and dt == L.BF16 and Hp % 256 == 0 and Zp <= 512 and
                   16 * (Hp + 8) * 2 + 16 * (Zp + 8) * 2 + 2 * 16 * Zp * 4 <= 128 * 1024)
        self.fused_q = bool(fused_q)
        if fused_q:
            wl1, ws1, wl2, ws2 = e.sh["q_location_w1"], e.sh["q_scale_w1"], e.sh["q_location_w2"], e.sh["q_scale_w2"]
            e._call(P, lib.vmmt_qnet_fwd, dt, ctx.p(), ctx.ld, self.src_len.data_ptr(), wl1.p(), ws1.p(), wl1.ld,
                    e.pp("inf_net_global.location.fc1.bias"), e.pp("inf_net_global.scale.fc1.bias"), wl2.p(), ws2.p(), wl2.ld,
                    e.pp("inf_net_global.location.fc2.bias"), e.pp("inf_net_global.scale.fc2.bias"), self.eps.p(), self.hbar.p(),
                    self.hbar.ld, self.q_h1["location"].p(), self.q_h1["scale"].p(), self.q_h1["location"].ld, self.mu.p(),
                    self.sigma.p(), self.z32.p(), self.zT.p(), self.zT.ld, self.kl_b.data_ptr(), self.stats.data_ptr(), B, S, Hp, Zp, Z,
                    1 if training else 0, 1 if e.qnet_split else 0)
            if e.qnet_split:        # the two networks ran in separate workgroups: the sample and the KL in a small launch of their own
                e._call(P, lib.vmmt_latent_fwd, dt, self.mu.p(), self.sigma.p(), self.eps.p(), self.z32.p(), self.zT.p(), self.zT.ld,
                        self.kl_b.data_ptr(), self.stats.data_ptr(), B, Z, 1 if training else 0)
        else:
            # a4 q(z|x): masked mean of the detached memory, two 2-layer MLPs
            if not d.conditional:
                e._call(P, lib.vmmt_masked_mean, dt, ctx.p(), ctx.ld, self.src_len.data_ptr(), self.hbar.p(), self.hbar.ld, B, S, H)
            # the two MLPs are independent and sit on the critical path between encoder and decoder (four latency-bound GEMMs):
            # the scale branch runs on the side stream (idle at this point) next to the location branch
            if not d.conditional:
                e._record(P, "hbar_ready")
            for br, outb, act in (() if d.conditional else (("location", self.mu, L.ACT_NONE), ("scale", self.sigma, L.ACT_SOFTPLUS))):
                if br == "scale" and e.q_parallel:
                    e._sid = SIDE
                    e._wait(P, "hbar_ready")
                w1, w2 = e.sh["q_%s_w1" % br], e.sh["q_%s_w2" % br]
                e._gemm(P, L.GEMM_NT, self.hbar.p(), self.hbar.ld, w1.p(), w1.ld, self.q_h1[br].p(), self.q_h1[br].ld, B, Z, H,
                        addend=e.pp("inf_net_global.%s.fc1.bias" % br), ld_add=Z, add_rows=1, act=L.ACT_RELU)
                e._gemm(P, L.GEMM_NT, self.q_h1[br].p(), self.q_h1[br].ld, w2.p(), w2.ld, outb.p(), outb.ld, B, Z, Z,
                        addend=e.pp("inf_net_global.%s.fc2.bias" % br), ld_add=Z, add_rows=1, act=act, out_f32=1)
                if br == "scale" and e.q_parallel:
                    e._record(P, "sigma_ready")
                    e._sid = MAIN
                    e._wait(P, "sigma_ready")
            # a5 fused mu/sigma -> sample -> KL
            if d.conditional:
                e._call(P, lib.vmmt_latent_cond_fwd, dt, self.mu.p(), self.sigma.p(), self.mu_p.p(), self.sigma_p.p(), self.eps.p(),
                        self.z32.p(), self.zT.p(), self.zT.ld, self.kl_b.data_ptr(), self.stats.data_ptr(), B, Z, 1 if training else 0)
            else:
                e._call(P, lib.vmmt_latent_fwd, dt, self.mu.p(), self.sigma.p(), self.eps.p(), self.z32.p(), self.zT.p(), self.zT.ld,
                        self.kl_b.data_ptr(), self.stats.data_ptr(), B, Z, 1 if training else 0)
        # a8 image network (location branch only; the scale branch is dead, H6 / VILoss.py:321): side stream, under the decoder
        e._record(P, "z_ready")
        e._sid = SIDE
        e._wait(P, "z_ready")
        if emb_tab is not None:
            # the [S B x E] copy of the source embeddings for the backward's dW_ih product only: here, behind a wait that orders this stream
            # behind the main stream's catch-up of the batch's rows, off every critical path (joined by `img_fwd`)
            src_gather()
        if training and self.z_in_Xt:       # z over the decoder steps next to the target embeddings (see Xt): for the backward's dW_ih
            e._call(P, lib.vmmt_gather_rows, dt, self.z32.p(), self.z32.ld, self.zrep_ids.data_ptr(), self.Xt.p(0, self.Ep), self.Xt.ld, M, Z)
        e._call(P, lib.vmmt_gate_fwd, dt, self.z32.p(), e.pp("inf_net_image.gate_affine_transform.weight"),
                e.pp("inf_net_image.gate_affine_transform.bias"), self.gate.data_ptr(), self.zt.p(), self.zt.ld, B, Z)
        w1, w2 = e.sh["iv_w1"], e.sh["iv_w2"]
        e._gemm(P, L.GEMM_NT, self.zt.p(), self.zt.ld, w1.p(), w1.ld, self.h1v.p(), self.h1v.ld, B, D, Z,
                addend=e.pp("inf_net_image.location.fc1.bias"), ld_add=D, add_rows=1, act=L.ACT_RELU)
        e._gemm(P, L.GEMM_NT, self.h1v.p(), self.h1v.ld, w2.p(), w2.ld, self.mu_v.p(), self.mu_v.ld, B, D, D,
                addend=e.pp("inf_net_image.location.fc2.bias"), ld_add=D, add_rows=1, out_f32=1)
        if early_head or aux_zero:
            e._wait(P, "aux_fwd")        # (join: whoever is behind the side stream's forward is behind the gradient zeroing as well)
        e._record(P, "img_fwd")
        e._sid = MAIN
        # a6 decoder: gx[t] = emb(y_t) W_e^T (side stream, above) ; zx = z W_z^T + b is added inside the step kernel
        e._wait(P, "opt_side_done")      # decoder / attention / generator parameters + shadows of the previous update
        wz, bsum = e.sh["dec_wih_l0_z"], e.sh["dec_b_l0"]
        e._gemm(P, L.GEMM_NT, self.zT.p(), self.zT.ld, wz.p(), wz.ld, self.zx.p(), self.zx.ld, B, 4 * Hp, Z,
                addend=bsum.p(), ld_add=bsum.ld, add_rows=1, out_f32=1)
        x, xcols = self.Xt, E
        for l in range(Lyr):
            if l == 0:
                e._wait(P, "dec_gx")
            else:
                wi, bs = e.sh["dec_wih_l%d" % l], e.sh["dec_b_l%d" % l]
                e._gemm(P, L.GEMM_NT, x.p(), x.ld, wi.p(), wi.ld, self.dec_gx[l].p(), self.dec_gx[l].ld, M, 4 * Hp, H,
                        addend=bs.p(), ld_add=bs.ld, add_rows=1, out_f32=1)
            last = l == Lyr - 1
            outb, ocol = (self.cat, Hp) if last else (self.dec_out[l], 0)
            whh = e.sh["dec_whh_l%d" % l]
            seq = (L.LstmDirFwd * Tp)()
            for t in range(Tp):
                a = seq[t]
                if t == 0:
                    a.h_prev, a.ld_hprev = self.hn[l].p(), self.hn[l].ld
                    a.c_prev, a.ld_cprev = self.cn[l].p(), self.cn[l].ld
                else:
                    a.h_prev, a.ld_hprev = outb.p((t - 1) * B, ocol), outb.ld
                    a.c_prev, a.ld_cprev = self.dec_c[l].p((t - 1) * B), self.dec_c[l].ld
                a.w_hh, a.ld_w = whh.p(), whh.ld
                a.gx, a.ld_gx = self.dec_gx[l].p(t * B), self.dec_gx[l].ld
                if l == 0:
                    a.gx2, a.ld_gx2 = self.zx.p(), self.zx.ld
                a.gates, a.ld_gates = self.dec_gates[l].p(t * B), self.dec_gates[l].ld
                a.c_out, a.ld_c = self.dec_c[l].p(t * B), self.dec_c[l].ld
                a.h_out, a.ld_h = outb.p(t * B, ocol), outb.ld
                a.h_n, a.c_n, a.t, a.capture = None, None, t, 0
            e._lstm_seq_fwd(P, seq, 1, Tp, None, B, Hp)
            if not last:
                x, xcols = self.dec_out[l], H
                if drop:
                    e._call(P, lib.vmmt_mul, dt, self.dec_out[l].p(), self.dec_out[l].ld, self.dec_mask[l].p(), self.dec_mask[l].ld,
                            self.dec_xdrop[l].p(), self.dec_xdrop[l].ld, M, H)
                    x = self.dec_xdrop[l]
        # a7 attention
        wa, wo = e.sh["wa"], e.sh["wo"]
        e._gemm(P, L.GEMM_NT, self.cat.p(0, Hp), self.cat.ld, wa.p(), wa.ld, self.Q.p(), self.Q.ld, M, H, H)
        e._call(P, lib.vmmt_attn_fwd, dt, self.Q.p(), self.Q.ld, ctx.p(), ctx.ld, self.src_len.data_ptr(), self.cat.p(), self.cat.ld,
                self.probs.data_ptr(), Tp, B, S, Hp)
        e._gemm(P, L.GEMM_NT, self.cat.p(), self.cat.ld, wo.p(), wo.ld, self.AH.p(), self.AH.ld, M, H, 2 * Hp, act=L.ACT_TANH)
        # join: the loss plans read mu_v.  ONE wait for everything the other streams did in this plan -- `img_fwd` is the side stream's last
        # record: the mask (`out_mask`, recorded before it on that stream, or on the aux stream which that stream joined: `aux_fwd`) and both
        # halves of the last optimiser step (issued on those streams in front of this plan's work) are behind it.  Every wait costs the main
        # stream ~6 us here, between the decoder and the sweep: three of them did (out_mask, img_fwd, opt_gen_done in the loss plan)
        e._wait(P, "img_fwd")
        if drop:
            e._call(P, lib.vmmt_mul, dt, self.AH.p(), self.AH.ld, self.out_mask.p(), self.out_mask.ld, self.O.p(), self.O.ld, M, H)
        # the ONE wait above stands for three that used to be explicit (out_mask, opt_gen_done, the gradient zeroing): checked here, so that
        # an edit that moves one of them to a stream `img_fwd` does not cover fails when the plan is built, not as a race
        if not d.conditional and e.use_side_stream:
            need = (["vmmt_zero_multi"] if training else []) + (["vmmt_dropout_mask"] if drop else []) + (["BG_FLUSH2"] if early_head else [])
            self._assert_joined(P, need, "img_fwd")
        return P

    @staticmethod
    def _assert_joined(P, needed, event):
        """walks a launch plan: `needed` (entry names) must all lie behind the record of `event` -- on the recording stream itself or on a
        stream it had joined (a wait for an event recorded there) by then -- and the main stream must wait for `event` afterwards"""
        have = {sid: set() for sid in range(4)}
        snap, recorded_at, main_waits = {}, None, False
        for k, (fn, args, name, _keep, sid) in enumerate(P):
            if fn is None:
                if name == "EV_RECORD":
                    snap[args] = set(have[sid])
                    if args == event:
                        recorded_at = k
                elif name == "EV_WAIT":
                    have[sid] |= snap.get(args, set())
                    if args == event and sid == 0 and recorded_at is not None:
                        main_waits = True
                else:
                    have[sid].add(name)
            else:
                have[sid].add(name)
        missing = [n for n in needed if n not in snap.get(event, set())]
        if recorded_at is None or missing or not main_waits:
            raise AssertionError("launch plan: %s is not behind `%s` (recorded: %s, main stream waits for it: %s) -- the main stream's one "
                                 "join would not cover it" % (missing, event, recorded_at is not None, main_waits))

    def _plan_loss(self, training):
        """forward part of NMTVIModel1LossCompute._compute_loss (VILoss.py:217-513): statistics only."""
        e, d, lib = self.e, self.e.d, self.e.lib
        P = []
        wg = e.sh["wg"]
        e._sid = 0
        # (the generator's third of the last optimiser step -- `opt_gen_done`, side or aux stream -- is behind the forward plan's last wait: see there)
        O = self.O if (training and d.dropout > 0) else self.AH      # eval: nn.Dropout is the identity
        if training:
            self._loss_patch = None
        if training and self.gen_fused:
            # statistics AND dO = dL/dO in one sweep of Wg + the kernel that folds its vocabulary slices (1 / normalization is patched into
            # the latter by loss_backward: argument 10); the softmax weights P and the per-slice scaled copies O'_s of O they leave
            # behind feed the dWg GEMM of the backward plan
            assert O.ld == _ru(d.hid, KPAD)
            Kp = _ru(d.hid, KPAD)
            self._sweep_entry = len(P)           # (Workspace.set_token_count patches M / rows / strides of these two per step)
            e._call(P, lib.vmmt_gen_fwd_dO, e.dt, wg.p(), wg.ld, wg.t.shape[0], e.pp("generator.0.bias"), O.p(), O.ld, self.y.data_ptr(), self.M, d.vt,
                    Kp, self.gen_ws.data_ptr(), self.tgt_logit.data_ptr(), self.gen_P.data_ptr(), self.gen_ldp, None)
            self._loss_patch = (len(P), 10)
            # the fold in two launches (Engine.split_combine): here what the dWg product waits for (c_s, O'_s) and the statistics; dO -- the
            # fold of the slices' 63 MB of partial accumulators, which only the main stream's backward chain reads -- is the backward plan's
            # first main-stream entry, BEHIND the point where the side stream's dWg starts (`bwd_begin`)
            self._combine_args = (e.dt, wg.p(), wg.ld, O.p(), O.ld, self.y.data_ptr(), self.M, d.vt, Kp, PAD, 0.0,
                                  self.gen_ws.data_ptr(), self.tgt_logit.data_ptr(), self.lse.data_ptr(), self.tok_nll.data_ptr(),
                                  self.gen_y32.data_ptr(), self.dO32.p(), self.dO32.ld, self.stats.data_ptr(), self.gen_cs.data_ptr(),
                                  self.gen_Os.data_ptr(), self.gen_Os.shape[2], self.gen_Os.shape[1] * self.gen_Os.shape[2], None)
            self._combine_split = bool(e.split_combine)
            e._call(P, lib.vmmt_gen_fwd_combine_stats if self._combine_split else lib.vmmt_gen_fwd_combine, *self._combine_args)
            return P
        e._call(P, lib.vmmt_gen_loss_fwd, e.dt, wg.p(), wg.ld, e.pp("generator.0.bias"), O.p(), O.ld, self.y.data_ptr(),
                self.M, d.vt, _ru(d.hid, KPAD), PAD, self.part_max.data_ptr(), self.part_sum.data_ptr(), None,
                self.tgt_logit.data_ptr(), self.lse.data_ptr(), self.tok_nll.data_ptr(), self.stats.data_ptr())
        return P

    # ---------------------------------------------------------------------------------------------- backward plan
    def kl_sum_ptr(self):
        """the KL sum the backward weighs against the free-bits margin: this rank's statistic, or under data parallelism its
        all-reduced copy (`kl_global`, filled by the KL_ALLREDUCE entry of the backward plan)"""
        if self.e.dp_on():
            return self.kl_global.data_ptr()
        return self.stats.data_ptr() + 4 * L.STAT_KL_SUM

    def _latent_bwd_args(self, batch_global, kl_mult, use_freebits, margin, inv_norm):
        rp = bool(self.e.reparam_grad) and hasattr(self, "dz")
        dzp, epp = (self.dz.p(), self.eps.p()) if rp else (None, None)
        if self.e.d.conditional:
            return (self.e.dt, self.mu.p(), self.sigma.p(), self.mu_p.p(), self.sigma_p.p(), self.kl_sum_ptr(),
                    float(batch_global), float(kl_mult), 1 if use_freebits else 0, float(margin), float(inv_norm),
                    dzp, epp, self.q_dmu.p(), self.q_dmu.ld, self.q_dpre.p(), self.q_dpre.ld, self.p_dmu.p(), self.p_dmu.ld,
                    self.p_dpre.p(), self.p_dpre.ld, self.B, self.e.d.z)
        return (self.e.dt, self.mu.p(), self.sigma.p(), self.kl_sum_ptr(), float(batch_global),
                float(kl_mult), 1 if use_freebits else 0, float(margin), float(inv_norm), dzp, epp, self.q_dmu.p(), self.q_dmu.ld,
                self.q_dpre.p(), self.q_dpre.ld, self.B, self.e.d.z)

    def set_token_count(self, n):
        """the generator's calls of THIS step run over the n decoder rows that carry a target (rounded up to 128; n = None or too close to
        T' B: all rows, as the reference does).  Patches the launch plans: token count, row map, the vocabulary-slice geometry that
        follows the count"""
        e, M = self.e, self.M
        Mc = M if n is None else min(M, _ru(max(int(n), 1), 128))
        if not (self.gen_fused and e.gen_compact) or Mc not in self.gen_geo:
            Mc = M
        if Mc == M:
            geo, rows, os_stride = (self.gen_ns, self.gen_vps, self.gen_mpad), None, self.gen_Os.shape[1] * self.gen_Os.shape[2]
        else:
            geo, rows, os_stride = self.gen_geo[Mc], self.gen_rows.data_ptr(), (Mc + KPAD) * self.gen_Os.shape[2]
        self.gen_Mc, self.gen_cur = Mc, (geo, rows, os_stride)
        if getattr(self, "_compact_entry", None) is not None:
            fn, a, name, keep, sid = self.plan_fwd_train[self._compact_entry]
            self.plan_fwd_train[self._compact_entry] = (fn, a[:3] + (Mc,) + a[4:], name, keep, sid)
        P = self.plan_loss_train
        ii = self._sweep_entry
        fn, a, name, keep, sid = P[ii]
        P[ii] = (fn, a[:8] + (Mc,) + a[9:15] + (rows,), name, keep, sid)
        fn, a, name, keep, sid = P[ii + 1]
        P[ii + 1] = (fn, a[:6] + (Mc,) + a[7:22] + (os_stride, rows), name, keep, sid)
        self._bwd_tokens = None               # the backward plan follows in backward_plan()

    def _patch_bwd_tokens(self):
        if self._bwd_tokens == self.gen_Mc or not self.gen_fused or not hasattr(self, "_dwg_gemm"):
            return
        (ns, vps, mpad), rows, os_stride = self.gen_cur
        a = self._dwg_gemm
        a.K, a.b_batch_rows, a.b_batch_stride = (self._dwg_K0 if self.gen_Mc == self.M else self.gen_Mc), vps, os_stride
        if a.colsum_w:
            a.colsum_w_stride = mpad
        P = self.plan_bwd
        fn, args, name, keep, sid = P[self._finish_entry]
        P[self._finish_entry] = (fn, args[:7] + (self.gen_Mc,) + args[8:15] + (rows,), name, keep, sid)
        if getattr(self, "_comb_dO_entry", None) is not None:          # (the second half of the fold: as the loss plan's entry, set_token_count)
            fn, args, name, keep, sid = P[self._comb_dO_entry]
            P[self._comb_dO_entry] = (fn, args[:6] + (self.gen_Mc,) + args[7:22] + (os_stride, rows), name, keep, sid)
        self._bwd_tokens = self.gen_Mc

    def backward_plan(self, inv_norm, batch_global, kl_mult, use_freebits, margin, drop):
        key = (bool(drop), bool(self.e.reparam_grad))      # the only STRUCTURAL inputs; every scalar is patched below
        if self._bwd_key != key:
            self.plan_bwd = self._plan_backward(inv_norm, batch_global, kl_mult, use_freebits, margin, drop)
            self._bwd_key = key
        P = self.plan_bwd
        self.e._kl_sum_needed = bool(use_freebits)
        fn, _, name, keep, sid = P[self._latent_bwd_index]
        P[self._latent_bwd_index] = (fn, self._latent_bwd_args(batch_global, kl_mult, use_freebits, margin, inv_norm), name, keep, sid)
        for ii, pos in self._patch.values():
            fn, args, name, keep, sid = P[ii]
            P[ii] = (fn, args[:pos] + (float(inv_norm),) + args[pos + 1:], name, keep, sid)
        if self.gen_fused and getattr(self, "_bwd_tokens", 0) != self.gen_Mc:
            self._patch_bwd_tokens()
        return P

    def ones_col(self):
        if not hasattr(self, "_ones"):
            self._ones = Buf(self.M, 1, self.e.T, self.e.dev, fill=1.0)
        return self._ones
