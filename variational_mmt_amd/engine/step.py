"""The step API of the engine: forward, loss, backward, statistics, optimiser (mixin of core.Engine)."""
import ctypes as C
import math
from os import environ as _os_env

import torch

from .. import _lib as L
from .layout import Buf, KPAD, PAD, SEG_ALIGN, _ru  # noqa: F401


class StepAPI(object):
    def set_image_table(self, table):
        """`table`: fp32 [N, D] image-feature array (numpy or tensor); kept resident in HBM
        (reference: host numpy + per-step fancy-index + H2D copy, TrainerMultimodal.py:632-639)."""
        t = torch.as_tensor(table)
        self.img_table = t.to(device=self.dev, dtype=torch.float32).contiguous()
        assert self.img_table.shape[1] == self.d.img

    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def forward(self, src, src_len, tgt, img_indices, training=True, eps=None, masks=None, table=None, tgt_len=None, n_tgt_tokens=None):
        """NMTVIModel.forward (Models.py:850-1011).  src [S,B] int64, src_len [B], tgt [T,B] (incl. <s>, </s>),
        img_indices [B] rows of the resident image table.  Returns the Workspace holding every activation."""
        S, B = int(src.shape[0]), int(src.shape[1])
        Tp = int(tgt.shape[0]) - 1
        if S > 256:
            raise RuntimeError("source length %d > 256 not supported by the attention kernels" % S)
        if not training:
            self.flush_lazy_rows()          # (evaluation plans look rows up without the lazy tables' mark / catch-up entries)
        self._in_step += 1
        try:
            return self._forward(src, src_len, tgt, img_indices, training, eps, masks, table, tgt_len, n_tgt_tokens, S, B, Tp)
        finally:
            self._in_step -= 1

    def _forward(self, src, src_len, tgt, img_indices, training, eps, masks, table, tgt_len, n_tgt_tokens, S, B, Tp):
        ws = self.workspace(B, S, Tp)
        st = self.stream()
        self.refresh_shadows(st)
        dev = self.dev
        tab = table if table is not None else getattr(self, "img_table", None)
        if tab is None:
            raise RuntimeError("no image-feature table: call set_image_table() first")
        d = self.d

        def dev64(t):
            t = torch.as_tensor(t)
            return t.to(device=dev, dtype=torch.int64, non_blocking=True).contiguous()
        src_d, tgt_d, len_d, idx_d = dev64(src), dev64(tgt), dev64(src_len).reshape(-1), dev64(img_indices).reshape(-1)
        gen_eps = training and eps is None
        self.rng_counter += 1
        L.check(self.lib.vmmt_prepare_batch(*((src_d.data_ptr(), tgt_d.data_ptr(), len_d.data_ptr(), idx_d.data_ptr(), S, Tp + 1, B,
                                            ws.S, ws.Tp + 1, PAD,
                                            ws.src.data_ptr(), ws.tgt_in.data_ptr(), ws.y.data_ptr(), ws.src_len.data_ptr(),
                                            ws.img_idx.data_ptr(), ws.stats.data_ptr(), ws.eps.p() if gen_eps else None,
                                            B * d.z if gen_eps else 0, self.rng_counter) + self._row_flag_args(training) + (st,))), "vmmt_prepare_batch")
        ws._inputs_keepalive = (src_d, tgt_d, len_d, idx_d)
        if d.conditional:
            if tgt_len is None:
                raise RuntimeError("the conditional model needs tgt_lengths (q(z|x,y,v) averages the target encodings)")
            ws.tgt_len.copy_(dev64(tgt_len).reshape(-1))
            # rows b*T + t (encoder_tgt sees the transposed target); positions beyond the batch's T hold the pad id
            ws.tgt_bt.fill_(PAD)
            ws.tgt_bt.view(B, ws.Tn)[:, :Tp + 1].copy_(tgt_d.reshape(Tp + 1, B).t())
        if training:
            if eps is not None:
                ws.eps.view().copy_(eps.to(device=dev, dtype=torch.float32))
            if d.dropout > 0:
                mk = [("enc_l%d" % l, ws.enc_mask[l]) for l in range(d.layers - 1)] + \
                     [("dec_l%d" % l, ws.dec_mask[l]) for l in range(d.layers - 1)]
                if d.conditional:
                    mk += [("enct_l%d" % l, ws.enct_mask[l]) for l in range(d.layers - 1)]
                for name, buf in mk:
                    if masks is not None and name in masks:
                        buf.view().copy_(masks[name].reshape(buf.rows, buf.cols).to(device=dev, dtype=self.T))
                    else:
                        self.rng_counter += 1
                        # the mask is generated over the padded buffer (pad columns are never read)
                        L.check(self.lib.vmmt_dropout_mask(self.dt, buf.p(), buf.rows * buf.ld, d.dropout, self.rng_counter, st),
                                "vmmt_dropout_mask")
                # the output mask is a plan entry on the side stream: give it this step's seed, or turn it into a no-op
                # when the caller injects the mask (tests)
                ii, buf = ws._mask_entries["dec_out"]
                fn, args, name, keep, sid = ws.plan_fwd_train[ii]
                self.rng_counter += 1
                if masks is not None and "dec_out" in masks:
                    buf.view().copy_(masks["dec_out"].reshape(buf.rows, buf.cols).to(device=dev, dtype=self.T))
                    ws.plan_fwd_train[ii] = (fn, (args[0], args[1], 0, args[3], self.rng_counter), name, keep, sid)
                else:
                    ws.plan_fwd_train[ii] = (fn, (args[0], args[1], buf.rows * buf.ld, args[3], self.rng_counter), name, keep, sid)
        if training and ws.gen_fused:
            # decoder rows that carry a target: told by the caller (a loader knows it) or counted here when the ids are still on the host
            n_tok = n_tgt_tokens
            if n_tok is None and self.gen_compact and torch.is_tensor(tgt) and not tgt.is_cuda:
                n_tok = int((tgt[1:] != PAD).sum())
            ws.set_token_count(n_tok)
        plan = ws.plan_fwd_train if training else ws.plan_fwd_eval
        ii = ws._img_idx[bool(training)]
        fn, args, name, keep, sid = plan[ii]
        plan[ii] = (fn, (L.F32, tab.data_ptr(), tab.shape[1]) + tuple(args[3:]), name, keep, sid)
        self._run(plan, ws.events)
        ws.training = training
        return ws

    def loss(self, ws):
        """statistics of _compute_loss without backward (monolithic_compute_loss, Loss.py:68-86)."""
        st = self.stream()
        self._run(ws.plan_loss_train if ws.training else ws.plan_loss_eval, ws.events)
        L.check(self.lib.vmmt_image_loss(self.dt, ws.mu_v.p(), ws.mu_v.ld, ws.img.p(), ws.img.ld, ws.B, self.d.img, 0.0, None, 0,
                                         ws.stats.data_ptr(), st), "vmmt_image_loss")
        return ws

    def loss_backward(self, ws, normalization=None, batch_global=None, kl_mult=1.0, use_freebits=False, margin=0.0,
                      zero_grad=True):
        """sharded_compute_loss (Loss.py:88-132): loss statistics + `loss.div(normalization).backward()`
        through the whole model into the gradient arena.  H3: all T' rows are used (monolithic semantics)."""
        st = self.stream()
        B = ws.B
        norm = float(normalization if normalization is not None else B)
        bg = float(batch_global if batch_global is not None else B)
        # the gradient arena was zeroed by the training forward plan (side stream); zero_grad=False is meaningless here
        if not ws.training:
            raise RuntimeError("loss_backward() after an eval-mode forward")
        self._in_step += 1
        try:
            if ws._loss_patch is not None:          # fused generator: the statistics pass writes dO = dL/dO scaled by 1 / normalization
                ii, pos = ws._loss_patch
                fn, args, name, keep, sid = ws.plan_loss_train[ii]
                ws.plan_loss_train[ii] = (fn, args[:pos] + (float(1.0 / norm),) + args[pos + 1:], name, keep, sid)
            self._run(ws.plan_loss_train, ws.events)
            plan = ws.backward_plan(1.0 / norm, bg, kl_mult, use_freebits, margin, bool(ws.training))
            self._cur_ws = ws
            self._run(plan, ws.events)
        finally:
            self._in_step -= 1
        return ws

    def read_stats(self, ws, batch_global=None, kl_mult=1.0, use_freebits=False, margin=0.0):
        """One D2H copy of the statistics vector -> the reference's loss_data dict (VILoss.py:483-497)."""
        s = ws.stats.tolist()
        B = float(batch_global if batch_global is not None else ws.B)
        kl_before = s[L.STAT_KL_SUM] / B
        kl_after = kl_before * kl_mult
        if use_freebits:
            kl_after = max(kl_after, margin)
        img_logprob = s[L.STAT_IMG_LOGPROB]
        nmt = s[L.STAT_NLL]
        return dict(nmt=nmt, td_kl_before=kl_before, td_kl_after=kl_after, td_kl_multiplier=kl_mult,
                    img_feats_loss=img_logprob, img_feats_cos=s[L.STAT_IMG_COS] / float(ws.B),
                    elbo=nmt - img_logprob + kl_after, n_words=int(round(s[L.STAT_NWORDS])),
                    n_correct=int(round(s[L.STAT_NCORRECT])))

    def optim_step(self, lr=0.002, max_grad_norm=5.0, beta1=0.9, beta2=0.999, eps=1e-9, grad_scale=1.0):
        """Optim.step (Optim.py:78-96): global-norm clip + Adam over the arena, then the compute shadows are refreshed.
        The arena is updated in two halves: [encoder | inference networks] on the current stream (the next forward needs
        them first), [generator | attention | decoder] on the side stream, where it overlaps the next step's encoder
        phase; the forward plan waits on `opt_side_done` before it touches decoder-side weights."""
        self._in_step += 1
        try:
            return self._optim_step(lr, max_grad_norm, beta1, beta2, eps, grad_scale)
        finally:
            self._in_step -= 1

    def _optim_step(self, lr, max_grad_norm, beta1, beta2, eps, grad_scale):
        main = torch.cuda.current_stream(self.dev)
        st = main.cuda_stream
        self._flush_bg()                # (two updates without a forward between them)
        self.poll_guard()               # a recurrence hand-off timed out a step or two ago: continue on the per-step kernels
        guard = self._guard.data_ptr()
        n_launch = [0]
        if self.dp_on() and self.dp.sharded:
            return self._optim_step_sharded(lr, max_grad_norm, beta1, beta2, eps, grad_scale)
        if self.dp_on():                # replicated data-parallel update: every rank must skip the same steps
            self.finish_allreduce()
            self.dp.all_reduce_tensor(self._guard[:1], "max")
        if max_grad_norm and not self._sumsq_by_plan:      # the backward plan normally accumulates the norm segment by segment
            self._sumsq[:L.SUMSQ_SLOTS].zero_()
            if self.rows_active():
                # row bookkeeping: rows flagged by EARLIER batches keep their old gradient (only the current batch's rows are cleared),
                # so the tables are normed by flagged row and everything between them densely -- the same pieces the plan entries take
                cur, slot = 0, 0
                for k, t in enumerate(sorted(self.row_tables, key=lambda t: t["off"])):
                    if t["off"] > cur:
                        L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * cur, t["off"] - cur, self._sumsq.data_ptr(), slot, st), "vmmt_sumsq")
                        slot += 1
                    L.check(self.lib.vmmt_sumsq_rows(self.flat_g.data_ptr() + 4 * t["off"], t["R"], t["C"], t["flags"].data_ptr(), t["hist"].data_ptr(),
                                                     t["rowsq"].data_ptr(), self._sumsq.data_ptr(), 3 + self.row_tables.index(t), st), "vmmt_sumsq_rows")
                    cur = t["end"]
                if self.n_opt > cur:
                    L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * cur, self.n_opt - cur, self._sumsq.data_ptr(), slot, st), "vmmt_sumsq")
            else:
                L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr(), self.n_opt, self._sumsq.data_ptr(), 0, st), "vmmt_sumsq")
        self._sumsq_by_plan = False
        self._step_count += 1
        split = self.offsets[self.first_enc_name][0]
        emb_fg = bool(self.d.conditional and self.cond_emb_fg)
        if emb_fg:
            # conditional model: the shared target embedding table (last item of the background half, no compute shadow) is updated in the
            # FOREGROUND: encoder_tgt's forward recurrence, the step's first long chain, gathers from it right at the start of the step
            split = self.offsets["decoder.embeddings.make_embedding.emb_luts.0.weight"][0]

        t_adam = self.step_count         # (by value: a held-back half of this update runs after optim_step has returned)

        def adam_range(lo, hi, stream, shadow=None):
            blocks = int(self.bg_adam_blocks) if stream != st else int(self.fg_adam_blocks)
            if hi > lo:
                L.check(self.lib.vmmt_adam_step(self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
                                                self.flat_m.data_ptr() + 4 * lo, self.flat_v.data_ptr() + 4 * lo, hi - lo, lr, beta1, beta2,
                                                eps, t_adam, float(max_grad_norm or 0.0), self._sumsq.data_ptr(), grad_scale,
                                                blocks, shadow, guard, stream), "vmmt_adam_step")
                n_launch[0] += 1

        rows = self.rows_active()
        if rows:
            if (beta1, beta2, eps) != self._adam_consts:
                # the replayed zero-gradient steps use the run's betas / eps: a change brings every row up to date under the old ones first
                self.flush_lazy_rows()
                self._adam_consts = (beta1, beta2, eps)
                self.drop_workspaces()          # (the catch-up entries of the forward plans carry them)
            self._lazy_dirty = True
        roll = int(self.lazy_roll)

        def rows_step(t, stream):
            o = 4 * t["off"]
            L.check(self.lib.vmmt_adam_rows_step(self.flat_p.data_ptr() + o, self.flat_g.data_ptr() + o, self.flat_m.data_ptr() + o,
                                                 self.flat_v.data_ptr() + o, t["R"], t["C"], t["flags"].data_ptr(), t["last"].data_ptr(),
                                                 t["hist"].data_ptr(), lr, beta1, beta2, eps, t_adam, roll, float(max_grad_norm or 0.0),
                                                 self._sumsq.data_ptr(), grad_scale, guard, stream),
                    "vmmt_adam_rows_step")
            n_launch[0] += 1

        def adam(lo, hi, stream):
            # the big unpadded bf16 shadows (generator weight, image network fc2) are written by the update itself: their range is
            # a launch of its own with the shadow attached, and the shadow refresh behind it skips them (_pack_tables); the embedding
            # tables are updated by the row-wise kernel (gradient read for the batch's rows only) when the row bookkeeping is on
            pieces = [(s_lo, s_hi, ("shadow", ptr)) for s_lo, s_hi, ptr in self._fused_shadows() if lo <= s_lo and s_hi <= hi]
            if rows:
                pieces += [(t["off"], t["end"], ("rows", t)) for t in self.row_tables if lo <= t["off"] and t["end"] <= hi]
            sh = [x for x in pieces if x[2][0] == "shadow"]
            tb = [x for x in pieces if x[2][0] == "rows"]
            one = int(self.adam_one_launch)
            if (one == 1 or (one == 2 and stream == st) or (one == 3 and stream != st)) and hi > lo and len(sh) <= 1 and len(tb) <= 1:
                # ONE dense launch for the whole range -- around the lazily updated table (a hole), the shadow attached to its piece -- and the
                # table's row-wise launch: the update's pieces are small, their launches were the step's tail (vmmt_adam_step_ranges)
                blocks = int(self.bg_adam_blocks) if stream != st else int(self.fg_adam_blocks)
                h_lo, h_n = (tb[0][0] - lo, tb[0][1] - tb[0][0]) if tb else (0, 0)
                s_ptr, s_lo, s_n = (sh[0][2][1], sh[0][0] - lo, sh[0][1] - sh[0][0]) if sh else (None, 0, 0)
                if hi - lo - h_n > 0:
                    L.check(self.lib.vmmt_adam_step_ranges(self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
                                                           self.flat_m.data_ptr() + 4 * lo, self.flat_v.data_ptr() + 4 * lo, hi - lo, h_lo, h_n, lr,
                                                           beta1, beta2, eps, t_adam, float(max_grad_norm or 0.0), self._sumsq.data_ptr(), grad_scale,
                                                           blocks, s_ptr, s_lo, s_n, guard, stream), "vmmt_adam_step_ranges")
                    n_launch[0] += 1
                if tb:
                    rows_step(tb[0][2][1], stream)
                return
            cur = lo
            for p_lo, p_hi, (kind, what) in sorted(pieces, key=lambda x: x[0]):
                adam_range(cur, p_lo, stream)
                if kind == "shadow":
                    adam_range(p_lo, p_hi, stream, what)
                else:
                    rows_step(what, stream)
                cur = p_hi
            adam_range(cur, hi, stream)
        if self.use_side_stream and self.split_optim:
            # both halves are HBM-bound: the critical half runs alone at full bandwidth, the other one starts behind it
            adam(split, self.n_opt, st)
            ev = self.global_events.setdefault("adam_main_done", torch.cuda.Event())
            ev.record(main)
            self._pack_part(2, st)
            side = self.side_stream

            defer = bool(self.hold_back and self.bg_after_head and not self.d.conditional and not self.dp_on())
            # held back (bg_after_head): [attention | decoder | target embeddings] and their shadows first -- `opt_side_done`, what the
            # decoder's forward waits for -- and the generator's third of the arena, which nothing reads before the vocabulary sweep, behind
            # the decoder's input projection (`opt_gen_done`)
            cut = self.segments[1][0] if (defer and self._bg_cut_ok()) else 0

            def part_a(stream=None):
                side.wait_event(ev)
                adam(cut, split, side.cuda_stream)
                if cut == 0:
                    self.global_events.setdefault("opt_gen_done", torch.cuda.Event()).record(side)
                self._pack_part(3, side.cuda_stream)
                self.global_events.setdefault("opt_side_done", torch.cuda.Event()).record(side)

            def part_b(stream=None):
                s2 = stream if stream is not None else side        # (the forward plan places the generator's part on its AUX stream)
                if s2 is not side:
                    s2.wait_event(self.global_events["opt_side_done"])
                if cut > 0:
                    adam(0, cut, s2.cuda_stream)
                    self.global_events.setdefault("opt_gen_done", torch.cuda.Event()).record(s2)
                self._publish_guard(s2)
                self._adam_launches = max(1, n_launch[0])
            if defer:
                self._pending_bg = [part_a, part_b]     # issued by the next forward plan (BG_FLUSH / BG_FLUSH2), or by whoever waits for them
            else:
                part_a()
                part_b()
        else:
            adam(0, self.n_opt, st)
            self._pack_part(2, st)
            self._pack_part(3, st)
            self._publish_guard(main)
        self._adam_launches = max(1, n_launch[0])
        self.shadows_dirty = False
        self._bg_open = bool(self.use_side_stream and self.split_optim)     # (a read of the arena from outside a step waits for the side stream's half)

    def _optim_step_sharded(self, lr, max_grad_norm, beta1, beta2, eps, grad_scale):
        """Data-parallel optimiser step with the state sharded over the ranks (dp.GradSync.sharded).  The backward plan has
        reduce-scattered every arena segment, so this rank holds the SUM of the gradients for its 1 / world of each segment
        (dp.shard).  Here: squared norm of the own shards (the deterministic reduction of vmmt_sumsq, one slot per segment) ->
        all-gather of the ranks' slot totals, added in rank order: every rank computes the same clip coefficient bit for bit ->
        clip + Adam on the own shards only (28 B/param of HBM traffic over 1 / world of the arena) -> all-gather of the updated
        parameters, segment by segment: [encoder | inference networks] in the foreground (the next forward starts with them),
        [generator | attention + decoder] on the side stream underneath the next step's encoder -> shadow refresh.
        The Adam moments of the other ranks' shards are not maintained here (dp.GradSync.gather_moments collects them for a
        checkpoint).  Same update as the replicated path: the reduced gradient of an element is the same sum wherever it is
        formed and the update is element-wise (only the norm is added up in another order); the replicas stay bit-identical
        (tests/test_gpu_dp_two_ranks.py)."""
        dp = self.dp
        main = torch.cuda.current_stream(self.dev)
        st = main.cuda_stream
        guard = self._guard.data_ptr()
        n_launch = [0]
        dp.timed_wait("gradient_wait", self.finish_allreduce)      # the reduce-scatters of the backward plan
        self._step_count += 1
        segs = self.segments
        own = [dp.shard(lo, hi) for lo, hi in segs]
        if len(self._normed) != len(segs):      # (normally the backward plan normed every segment's shard behind its reduce-scatter)
            self._sumsq[:L.SUMSQ_SLOTS].zero_()
            for i, (a, b) in enumerate(own):
                if b > a:
                    L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * a, b - a, self._sumsq.data_ptr(), i, st), "vmmt_sumsq")
        self._sumsq_by_plan = False
        self._normed = set()
        # ONE 36-byte all-gather carries the ranks' norm partials and their guard words (a rank whose recurrence timed out must not be
        # the only one that skips the update: the replicas would part); folded in a fixed order on every rank (csrc/optim.hip)
        if getattr(self, "_dp_row", None) is None or self._dp_rows.shape[0] != dp.world:
            self._dp_row = torch.zeros(L.SUMSQ_SLOTS + 1, dtype=torch.float32, device=self.dev)
            self._dp_rows = torch.zeros(dp.world, L.SUMSQ_SLOTS + 1, dtype=torch.float32, device=self.dev)
        L.check(self.lib.vmmt_dp_norm_pack(self._sumsq.data_ptr(), guard, self._dp_row.data_ptr(), st), "vmmt_dp_norm_pack")
        dp.timed_wait("norm_all_gather", lambda: dp.all_gather_row_into(self._dp_rows, self._dp_row))
        L.check(self.lib.vmmt_dp_norm_fold(self._dp_rows.data_ptr(), dp.world, self._sumsq.data_ptr(), guard, st), "vmmt_dp_norm_fold")

        def adam(a, b, stream):
            if b > a:
                L.check(self.lib.vmmt_adam_step(self.flat_p.data_ptr() + 4 * a, self.flat_g.data_ptr() + 4 * a, self.flat_m.data_ptr() + 4 * a,
                                                self.flat_v.data_ptr() + 4 * a, b - a, lr, beta1, beta2, eps, self.step_count,
                                                float(max_grad_norm or 0.0), self._sumsq.data_ptr(), grad_scale,
                                                int(self.bg_adam_blocks) if stream != st else 0, None, guard, stream),
                        "vmmt_adam_step")
                n_launch[0] += 1
        fg, bg = (2, 3), (0, 1)                           # foreground: encoder + inference networks; background: generator, decoder
        for i in fg:
            adam(own[i][0], own[i][1], st)
        dp.timed_wait("param_all_gather_foreground", lambda: [dp.all_gather(self.flat_p, *segs[i]) for i in fg])
        self._pack_part(0, st)
        ev = self.global_events.setdefault("adam_main_done", torch.cuda.Event())
        ev.record(main)
        side = self.side_stream if (self.use_side_stream and self.split_optim) else main
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for i in bg:
                adam(own[i][0], own[i][1], side.cuda_stream)
            for i in bg:
                dp.all_gather(self.flat_p, *segs[i])
            self._pack_part(1, side.cuda_stream)
        if side is not main:
            self.global_events.setdefault("opt_side_done", torch.cuda.Event()).record(side)
        self._publish_guard(side)
        self._adam_launches = max(1, n_launch[0])
        self.shadows_dirty = False
        self._bg_open = side is not main
