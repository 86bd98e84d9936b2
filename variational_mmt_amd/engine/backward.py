"""The backward launch plan of a workspace (mixin of workspace.Workspace)."""
import ctypes as C
import math
from os import environ as _os_env

import torch

from .. import _lib as L
from .layout import Buf, KPAD, PAD, _ru  # noqa: F401


class BackwardPlan(object):
    def _plan_backward(self, inv_norm, batch_global, kl_mult, use_freebits, margin, training_dropout):
        """Backward of loss/normalization through the whole model.  Stream 0 carries the critical path
        (G^T -> dO -> attention -> decoder LSTM -> encoder LSTM); stream 1 carries everything that only produces
        parameter gradients (image / q(z|x) networks, dWg, dW_o, dW_a, LSTM weight/bias gradients, embedding scatters)."""
        e, d, lib = self.e, self.e.d, self.e.lib
        B, S, Tp, M, MS = self.B, self.S, self.Tp, self.M, self.MS
        H, Hd, E, Z, D, V, Lyr, dirs = d.hid, d.hd, d.emb, d.z, d.img, d.vt, d.layers, d.dirs
        Hp, Hdp = d.hp, d.hdp
        gmap_d, gmap_e = (Hp, H), (Hdp, Hd)       # padded gate blocks -> nn.LSTM's [4H] rows (vmmt_gemm_args.c_row_blk)
        dt = e.dt
        P = []
        drop = training_dropout and d.dropout > 0
        wg = e.sh["wg"]
        MAIN, SIDE, AUX = 0, 1, 2
        e._sid = MAIN
        e._record(P, "bwd_begin")
        # ================= main: generator backward seed G^T, dO = G Wg ================================================
        e._sid = MAIN
        # (the bias gradient = row sums of G^T comes out of the same kernel; the gradient arena was zeroed by the forward plan on its side or
        #  aux stream, behind the main stream's last join there: Workspace._assert_joined)
        fuse_db = True
        # entries that carry run-time scalars (1 / normalization, KL weights): patched per step by backward_plan(), so that
        # token normalisation (a different value every batch) does not rebuild the plan
        # (walking the vocabulary in 2-6 chunks, so that a chunk of G^T is consumed by dO / dWg while it is still in the Infinity
        #  Cache, was measured with tools/ab.py: 2.227-2.97 ms against 2.213 ms in one pass -- not kept)
        self._patch = {}
        self._src_rows_normed = False
        if self.gen_fused:
            pass        # dO32 came out of the loss plan (vmmt_gen_fwd_dO); dWg / db: one GEMM + vmmt_gen_dW_finish on the side stream below
        elif fuse_db:
            self._patch["gen"] = (len(P), 12)
            e._call(P, lib.vmmt_gen_loss_bwd_db, dt, wg.p(), wg.ld, e.pp("generator.0.bias"), self.O.p(), self.O.ld, self.y.data_ptr(),
                    M, V, _ru(H, KPAD), PAD, self.lse.data_ptr(), inv_norm, self.GT.p(), self.GT.ld, e.gp("generator.0.bias"), 0)
        else:
            self._patch["gen"] = (len(P), 12)
            e._call(P, lib.vmmt_gen_loss_bwd, dt, wg.p(), wg.ld, e.pp("generator.0.bias"), self.O.p(), self.O.ld, self.y.data_ptr(),
                    M, V, _ru(H, KPAD), PAD, self.lse.data_ptr(), inv_norm, self.GT.p(), self.GT.ld)
        if not self.gen_fused:
            e._record(P, "GT")
            e._gemm(P, L.GEMM_TN, self.GT.p(), self.GT.ld, wg.p(), wg.ld, self.dO32.p(), self.dO32.ld, M, H, V, out_f32=1,
                    split_k=max(1, min(8, (1024 * 128 * 128) // max(1, M * H))))
        # ================= aux: image term + its network (z is detached: independent of the text path) ==========
        rp = bool(e.reparam_grad)
        TGT = 3
        # the decoder's parameter gradients and the first-half norm go to the AUX stream, which is idle once its own chain (image /
        # q(z|x) networks) is through: behind the generator's products on the side stream they reached into the step's tail
        dec_on_aux = bool(e.dec_grads_on_aux and not rp and not d.conditional and e.use_aux_stream)
        # tail_norm_first: every piece of the gradient norm right behind the launch that completes its gradients, on that launch's stream,
        # instead of as two chains of three norms behind the step's last products (the update needs all of them: 60 us of tail)
        # (one-layer models: config 2 1.436-1.443 against 1.453-1.459 ms; the scripts' two layers measured 2.180-2.187 against 2.168-2.180 with the
        #  pieces spread and level with only the main stream's two norms moved: they keep that form)
        spread = bool(e.tail_norm_first and not e.dp_on() and dec_on_aux and d.layers == 1)
        self._inf_normed = self._tgt_rows_normed = self._enc_dense_normed = False
        def kl_and_q_backward():
            # --- KL term -> q(z|x) networks (mu, sigma receive gradient only through the KL: H2) ------------------
            # data parallelism: free bits compares the GLOBAL batch-mean KL with the margin (VILoss.py:463-476), so the KL sum is
            # all-reduced (one float, on this stream, long after the forward produced it) before the latent backward reads it
            P.append((None, None, "KL_ALLREDUCE", None, e._sid))
            self._latent_bwd_index = len(P)
            if d.conditional:
                e._call(P, lib.vmmt_latent_cond_bwd, *self._latent_bwd_args(batch_global, kl_mult, use_freebits, margin, inv_norm))
            else:
                e._call(P, lib.vmmt_latent_bwd, *self._latent_bwd_args(batch_global, kl_mult, use_freebits, margin, inv_norm))
            qx = self.hq if d.conditional else self.hbar        # input of the q network's first layer
            branches = (("location", self.q_dmu), ("scale", self.q_dpre))

            def data_grads(i, br, dy):      # d h1 = relu'(.) (dy W2); conditional: d h_y = d h_q[:, H:2H] (h_x detached, v is data)
                w2q = e.sh["q_%s_w2" % br]
                e._gemm(P, L.GEMM_NN, dy.p(), dy.ld, w2q.p(), w2q.ld, self.q_dh1[br].p(), self.q_dh1[br].ld, B, Z, Z)
                e._call(P, lib.vmmt_act_bwd, dt, L.ACT_RELU, self.q_dh1[br].p(), self.q_dh1[br].ld, 0, self.q_h1[br].p(), self.q_h1[br].ld,
                        None, 0, self.q_dh1[br].p(), self.q_dh1[br].ld, B, Z)
                if d.conditional:           # the h_y columns of W1: [H, 2H) as stored, [Hp, Hp + 2 htp) as computed
                    w1q = e.sh["q_%s_w1" % br]
                    e._gemm(P, L.GEMM_NN, self.q_dh1[br].p(), self.q_dh1[br].ld, w1q.p(0, Hp), w1q.ld, self.dhy.p(), self.dhy.ld, B, 2 * d.htp, Z,
                            accumulate=1 if i else 0)

            def weight_grads(part, br, dy):
                pre = "inf_net_global.%s" % br
                if part == 2:
                    e._gemm(P, L.GEMM_TN, dy.p(), dy.ld, self.q_h1[br].p(), self.q_h1[br].ld, e.gp(pre + ".fc2.weight"), Z, Z, Z, B, out_f32=1, split_k=-1)
                    e._call(P, lib.vmmt_colsum, dt, dy.p(), dy.ld, B, Z, 0, 0, e.gp(pre + ".fc2.bias"), None)
                else:
                    if d.conditional and d.pad:
                        # dW1 piece by piece: the input's column ranges are padded each to its own width (h_x : Hp, h_y : 2 x htp, v)
                        for c0, nc, co, cm in ((0, H, 0, None), (H, 2 * d.htp, Hp, (d.htp, d.ht)), (2 * H, D, Hp + 2 * d.htp, None)):
                            e._gemm(P, L.GEMM_TN, self.q_dh1[br].p(), self.q_dh1[br].ld, qx.p(0, co), qx.ld, e.gp(pre + ".fc1.weight", 0, c0), d.qin,
                                    Z, nc, B, out_f32=1, split_k=-1, cmap=cm)
                    else:
                        e._gemm(P, L.GEMM_TN, self.q_dh1[br].p(), self.q_dh1[br].ld, qx.p(), qx.ld, e.gp(pre + ".fc1.weight"), d.qin,
                                Z, d.qin, B, out_f32=1, split_k=-1)
                    e._call(P, lib.vmmt_colsum, dt, self.q_dh1[br].p(), self.q_dh1[br].ld, B, Z, 0, 0, e.gp(pre + ".fc1.bias"), None)

            if cond_first:
                # d h_y first (6 small kernels), then encoder_tgt's recurrence on its own stream; the weight gradients, p's backward
                # and the image network follow on this stream, next to the recurrence.  encoder_tgt scatters into the shared
                # target-embedding gradient, so the first arena half is finished behind it (finish_first_half at the end of the plan)
                for i, (br, dy) in enumerate(branches):
                    data_grads(i, br, dy)
                e._record(P, "dhy")
                e._sid = TGT
                e._wait(P, "dhy")
                self._cond_backward_tgt(P, drop)
                e._record(P, "tgt_done")
                e._sid = AUX
                self._cond_backward(P, drop)        # d h_x for the encoder chain of the main stream (event dhbar_p)
                for i, (br, dy) in enumerate(branches):
                    weight_grads(2, br, dy)
                    weight_grads(1, br, dy)
                return                              # aux_chain() closes the stream (all-reduce, aux_done) behind the image network
            if d.conditional:
                self._cond_backward(P, drop)
            for i, (br, dy) in enumerate(branches):
                weight_grads(2, br, dy)
                data_grads(i, br, dy)
                weight_grads(1, br, dy)
            if kl_first:
                return                              # aux_chain() closes the stream behind the image network
            e._allreduce(P, "inf_net_image.location.fc2.weight", None)     # inference networks: tail of the arena, done first
            if d.conditional:
                self._cond_backward_tgt(P, drop)
            e._record(P, "aux_done")

        # conditional model: the critical path of the whole backward is d h_y -> encoder_tgt's 2 x B-step recurrence -> its parameter
        # gradients, so the few kernels that produce d h_y go out first and the recurrence gets a stream of its own (TGT); everything
        # else of this chain (image network, weight gradients of q / p, p's backward) runs next to it
        cond_first = bool(d.conditional and not rp)
        # fixed prior: the KL / q(z|x) backward needs mu, sigma and the KL sum only -- it goes first, gated by the sample and by the forward
        # plan's gradient zeroing (`grad_zero`; no wait when that ran on this very stream), the image term follows when the forward's image
        # network is through
        kl_first = bool(e.aux_early and e.aux_kl_first and not d.conditional and not rp)
        def aux_chain():
            e._sid = AUX
            if kl_first:
                e._wait(P, "z_ready")
                if getattr(self, "_zero_sid", None) != AUX:
                    e._wait(P, "grad_zero")
                kl_and_q_backward()
            # the image term and the KL / q(z|x) backward depend on the forward only (mu_v, mu / sigma, the KL sum), not on the generator
            # loss: gated by the forward's image network (behind the step's gradient zeroing on the same stream) they start while the
            # decoder's forward is still running -- the host is a step ahead of the GPU, so the launches are already queued
            early_cond = bool(cond_first and e.aux_early and e.cond_aux_early)
            if early_cond:
                # conditional model: the chain d h_y -> encoder_tgt's backward recurrence is the critical path of the whole backward and
                # depends on the forward only (KL of q against p(z|x)): it starts behind the sample, 0.5 ms before the loss is through
                e._wait(P, "z_ready")
                e._wait(P, "grad_zero")
                kl_and_q_backward()
                e._wait(P, "img_fwd")
            else:
                e._wait(P, "img_fwd" if (e.aux_early and not d.conditional and not rp) else "bwd_begin")
                if cond_first:
                    kl_and_q_backward()
            self._patch["img"] = (len(P), 7)
            e._call(P, lib.vmmt_image_loss, dt, self.mu_v.p(), self.mu_v.ld, self.img.p(), self.img.ld, B, D, inv_norm,
                    self.dmu_v.p(), self.dmu_v.ld, self.stats.data_ptr())
            w1, w2 = e.sh["iv_w1"], e.sh["iv_w2"]
            e._gemm(P, L.GEMM_TN, self.dmu_v.p(), self.dmu_v.ld, self.h1v.p(), self.h1v.ld, e.gp("inf_net_image.location.fc2.weight"), D,
                    D, D, B, out_f32=1, split_k=-1)
            e._call(P, lib.vmmt_colsum, dt, self.dmu_v.p(), self.dmu_v.ld, B, D, 0, 0, e.gp("inf_net_image.location.fc2.bias"), None)
            # [B x D] x [D x D] with B = a few hundred rows: 32 tiles of 128 x 128 would leave 7/8 of the chip idle for 100 us, so the
            # reduction is split over workgroups (f32 atomics into dh1v32, zeroed with the gradient arena) and the ReLU backward reads f32
            e._gemm(P, L.GEMM_NN, self.dmu_v.p(), self.dmu_v.ld, w2.p(), w2.ld, self.dh1v32.p(), self.dh1v32.ld, B, D, D, out_f32=1,
                    split_k=max(1, min(D // 256, 512 // max(1, ((B + 127) // 128) * ((D + 127) // 128)))))
            e._call(P, lib.vmmt_act_bwd, dt, L.ACT_RELU, self.dh1v32.p(), self.dh1v32.ld, 1, self.h1v.p(), self.h1v.ld, None, 0,
                    self.dh1v.p(), self.dh1v.ld, B, D)
            e._gemm(P, L.GEMM_TN, self.dh1v.p(), self.dh1v.ld, self.zt.p(), self.zt.ld, e.gp("inf_net_image.location.fc1.weight"), Z,
                    D, Z, B, out_f32=1, split_k=-1)
            e._call(P, lib.vmmt_colsum, dt, self.dh1v.p(), self.dh1v.ld, B, D, 0, 0, e.gp("inf_net_image.location.fc1.bias"), None)
            e._gemm(P, L.GEMM_NN, self.dh1v.p(), self.dh1v.ld, w1.p(), w1.ld, self.dzt.p(), self.dzt.ld, B, Z, D, out_f32=1,
                    split_k=max(1, min(D // 256, 256 // max(1, ((B + 63) // 64) * ((Z + 63) // 64)))))
            e._call(P, lib.vmmt_gate_bwd, self.dzt.p(), self.dzt.ld, self.z32.p(), self.gate.data_ptr(),
                    e.gp("inf_net_image.gate_affine_transform.weight"), e.gp("inf_net_image.gate_affine_transform.bias"), B, Z)
            if rp and not hasattr(self, "dzrow"):
                self.dzrow = Buf(M, Z, torch.float32, e.dev)       # dgates_t W_z per decoder row
                self.dz = Buf(B, Z, torch.float32, e.dev, ld=Z)    # dL/dz of the reparameterised sample

            if cond_first:
                e._wait(P, "tgt_done")
                e._allreduce(P, "inf_net_image.location.fc2.weight", None)     # inference networks: tail of the arena, done first
                e._record(P, "aux_done")
            elif kl_first:
                e._allreduce(P, "inf_net_image.location.fc2.weight", None)     # inference networks: tail of the arena, done first
                e._record(P, "aux_done")
                if spread:
                    e._sumsq_entry(P, "inf_net_image.location.fc2.weight", None, 2)
                    self._inf_normed = True
            elif not rp:
                kl_and_q_backward()

        self._comb_dO_entry = None

        def main_head():
            # main: [dO: the second half of the sweep's fold,] dropout + tanh backward, linear_out, attention
            e._sid = MAIN
            if self.gen_fused and getattr(self, "_combine_split", False):
                self._comb_dO_entry = len(P)
                self._patch["comb_dO"] = (len(P), 10)
                e._call(P, lib.vmmt_gen_fwd_combine_dO, *self._combine_args)
            e._call(P, lib.vmmt_act_bwd, dt, L.ACT_TANH, self.dO32.p(), self.dO32.ld, 1, self.AH.p(), self.AH.ld,
                    self.out_mask.p() if drop else None, self.out_mask.ld if drop else 0, self.dPre.p(), self.dPre.ld, M, H)
            e._record(P, "dPre")
            wo, wa = e.sh["wo"], e.sh["wa"]
            e._gemm(P, L.GEMM_NN, self.dPre.p(), self.dPre.ld, wo.p(), wo.ld, self.dcat.p(), self.dcat.ld, M, 2 * Hp, H)
            ctx = self.enc_out[Lyr - 1]
            if S > 64 or Tp > 64:
                # longer than the per-sentence kernels take (S <= 64, T' <= 64): one wave per query / per source position
                if not hasattr(self, "attn_dots"):
                    self.attn_dots = torch.zeros(M, dtype=torch.float32, device=e.dev)
                e._call(P, lib.vmmt_attn_bwd_long, dt, self.dcat.p(), self.dcat.ld, self.probs.data_ptr(), self.Q.p(), self.Q.ld, ctx.p(),
                        ctx.ld, self.src_len.data_ptr(), self.dQ.p(), self.dQ.ld, self.dctx.p(), self.dctx.ld, Tp, B, S, Hp,
                        self.attn_dots.data_ptr())
            else:
                e._call(P, lib.vmmt_attn_bwd, dt, self.dcat.p(), self.dcat.ld, self.probs.data_ptr(), self.Q.p(), self.Q.ld, ctx.p(), ctx.ld,
                        self.src_len.data_ptr(), self.dQ.p(), self.dQ.ld, self.dctx.p(), self.dctx.ld, Tp, B, S, Hp)
            e._record(P, "dQ")
            e._gemm(P, L.GEMM_NN, self.dQ.p(), self.dQ.ld, wa.p(), wa.ld, self.dR.p(), self.dR.ld, M, H, H,
                    addend=self.dcat.p(0, Hp), ld_add=self.dcat.ld, add_rows=-1, add_is_T=1)

        # issue order = the order of this list.  With the fused generator dO exists when the plan starts, so the main stream's first
        # kernels go out first instead of behind the ~20 small launches of the aux chain (tools/ab.py: 2.009 against 2.037 ms, equal
        # in a second run); the conditional model keeps the aux chain first, it IS the critical path there (4.42 against 4.45 ms)
        main_first = bool(self.gen_fused and not d.conditional and e.bwd_main_first)
        # data parallelism: the backend runs every collective of a process group on ONE stream, in the order the HOST issued them.  The
        # KL all-reduce at the head of the aux chain is due ~0.3 ms into the step, the generator segment's reduce-scatter only when the
        # dWg product is through (~1.2 ms): issued in that order the one float waited for the 60 MB behind the product, and the aux
        # chain -- gated by the KL sum -- moved from underneath the decoder's forward into the backward recurrences (world-1 RCCL
        # rehearsal: 2.46 against 1.75 ms per step).  So with ranks attached the aux chain is issued BEFORE the side stream's products:
        # collectives then reach the backend in the order their inputs become ready (KL, inference networks, generator, decoder, encoder)
        aux_before_side = bool(main_first and e.dp_on())
        if main_first:
            main_head()
            if aux_before_side:
                aux_chain()
        else:
            aux_chain()
        # side: dWg = G^T O as soon as G^T exists, underneath dO = G Wg of the main stream (measured, tools/ab.py: issuing it behind dR,
        # underneath the LSTM backward chains instead, is 3 % slower -- 2.333 vs 2.264 ms)
        e._sid = SIDE
        e._wait(P, "bwd_begin")
        def gen_dw():
            # dWg[slice s] = P[:, slice s]^T O'_s: ONE plain GEMM (K = tokens; the B operand switches with the vocabulary slice), then the
            # bias gradient and the one-hot term
            Kp = _ru(H, KPAD)
            # the bias gradient's weighted column sums of P ride in the GEMM's first column tile where its large-tile path runs
            # (e.gen_db_in_gemm: switch for tools/ab.py); otherwise vmmt_gen_dW_finish makes its own pass over P
            fused_db = e._gemm(P, L.GEMM_TN, self.gen_P.data_ptr(), self.gen_ldp, self.gen_Os.data_ptr(), Kp, e.gp("generator.0.weight"), H,
                               V, H, M, out_f32=1, b_batch_rows=self.gen_vps, b_batch_stride=self.gen_Os.shape[1] * Kp,
                               colsum=(self.gen_cs.data_ptr(), self.gen_mpad, e.gp("generator.0.bias")) if e.gen_db_in_gemm else None)
            self._dwg_gemm = P[-1][3]           # its GemmArgs: reduction length / slice geometry follow the step's token count (set_token_count)
            self._dwg_K0 = self._dwg_gemm.K
            self._bwd_tokens = M
            Og = self.O if (training_dropout and d.dropout > 0) else self.AH
            self._patch["gen"] = (len(P), 10)
            # (bias gradient + one-hot term at the END of the side stream or of the aux stream instead: 1.988 / 1.967 against 1.931-1.934 ms)
            self._finish_entry = len(P)
            e._call(P, lib.vmmt_gen_dW_finish, dt, self.gen_P.data_ptr(), self.gen_ldp, self.gen_cs.data_ptr(), Og.p(), Og.ld,
                    self.gen_y32.data_ptr(), M, V, Kp, inv_norm, e.gp("generator.0.weight"), H, e.gp("generator.0.bias"), 1 if fused_db else 0,
                    None)
        if self.gen_fused:
            # right away: issued later (on the aux stream behind the image / q(z|x) backward, i.e. underneath the LSTM backward chains) the
            # GEMM competes with the persistent LSTM kernels for CUs -- 2.20 against 2.11-2.13 ms per step (tools/ab.py, arms on shared streams)
            gen_dw()
        else:
            e._wait(P, "GT")
            e._gemm(P, L.GEMM_NN, self.GT.p(), self.GT.ld, self.O.p(), self.O.ld, e.gp("generator.0.weight"), H, V, H, M, out_f32=1)   # plain store
            if not fuse_db:
                e._call(P, lib.vmmt_rowsum, dt, self.GT.p(), self.GT.ld, V, M, e.gp("generator.0.bias"))
        e._allreduce(P, "generator.0.weight", "decoder.attn.linear_out.weight")
        # the generator's share of the gradient norm (a quarter of the arena) right behind its product, on this stream, instead of in the
        # norm of the whole first half at the end of the decoder's parameter gradients (slot 7: free outside the conditional model)
        gen_norm_early = bool(not d.conditional)
        if gen_norm_early:
            e._sumsq_entry(P, "generator.0.weight", "decoder.attn.linear_out.weight", 7)
        if main_first:
            if not aux_before_side:
                aux_chain()
        else:
            main_head()
        e._sid = SIDE
        grp = [] if (e.group_wgrads and dt == L.BF16) else None          # the two attention weight gradients: one grid
        e._wait(P, "dPre")
        e._gemm(P, L.GEMM_TN, self.dPre.p(), self.dPre.ld, self.cat.p(), self.cat.ld, e.gp("decoder.attn.linear_out.weight"), 2 * H,
                H, 2 * Hp, M, out_f32=1, split_k=-1, cmap=(Hp, H), group=grp)
        e._wait(P, "dQ")
        e._gemm(P, L.GEMM_TN, self.dQ.p(), self.dQ.ld, self.cat.p(0, Hp), self.cat.ld, e.gp("decoder.attn.linear_in.weight"), H,
                H, H, M, out_f32=1, split_k=-1, group=grp)
        e._gemm_group(P, grp)
        # (the decoder's weight gradients held back until the encoder's backward recurrence is through -- both backward recurrences
        #  without guests, everything in the tail -- was measured in round 4: 1.80 against 1.735 ms per step; LABNOTES.md)
        # ================= decoder LSTM backward (main) + its parameter gradients (side / aux) ====================
        dh_above = self.dR
        for l in reversed(range(Lyr)):
            e._sid = MAIN
            last = l == Lyr - 1
            outb, ocol = (self.cat, Hp) if last else (self.dec_out[l], 0)
            whhT = e.sh["dec_whhT_l%d" % l]
            dg = self.dec_dgates[l]
            seq = (L.LstmDirBwd * (Tp + 1))()                   # Tp cell-backward steps + the dh0 step
            for i, t in enumerate(reversed(range(Tp))):
                a = seq[i]
                if t < Tp - 1:
                    a.dgates_next, a.ld_dgn = dg.p((t + 1) * B), dg.ld
                a.w_hh_t, a.ld_wt = whhT.p(), whhT.ld
                a.dh_above, a.ld_dha = dh_above.p(t * B), dh_above.ld
                a.gates, a.ld_gates = self.dec_gates[l].p(t * B), self.dec_gates[l].ld
                a.c_t, a.ld_ct = self.dec_c[l].p(t * B), self.dec_c[l].ld
                if t > 0:
                    a.c_prev, a.ld_cp = self.dec_c[l].p((t - 1) * B), self.dec_c[l].ld
                else:
                    a.c_prev, a.ld_cp = self.cn[l].p(), self.cn[l].ld
                a.dc_carry, a.ld_dcc = self.dec_dcc[l].p(), self.dec_dcc[l].ld
                a.dgates_out, a.ld_dgo = dg.p(t * B), dg.ld
                a.t, a.inject = t, 0
            a = seq[Tp]                                        # gradient of the initial hidden state: dgates_0 W_hh (mode 1)
            a.dgates_next, a.ld_dgn = dg.p(0), dg.ld
            a.w_hh_t, a.ld_wt = whhT.p(), whhT.ld
            a.dh0_out, a.ld_dh0 = self.dec_dh0[l].p(), self.dec_dh0[l].ld
            # (conditional model: encoder_tgt's persistent backward is running by now; two persistent launches share a CU only if their
            #  registers fit one SIMD file and their LDS one CU -- 272 + 166 registers, 72 + 40 KiB here)
            e._lstm_seq_bwd(P, seq, 1, Tp, None, B, Hp, with_dh0=1)
            e._record(P, "dec_dg%d" % l)
            if l > 0:       # gradient w.r.t. the layer input stays on the critical path
                wi = e.sh["dec_wih_l%d" % l]
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, wi.p(), wi.ld, self.dec_dx[l - 1].p(), self.dec_dx[l - 1].ld, M, H, 4 * Hp)
                if drop:
                    e._call(P, lib.vmmt_mul, dt, self.dec_dx[l - 1].p(), self.dec_dx[l - 1].ld, self.dec_mask[l - 1].p(),
                            self.dec_mask[l - 1].ld, self.dec_dx[l - 1].p(), self.dec_dx[l - 1].ld, M, H)
                dh_above = self.dec_dx[l - 1]
            # ---- side / aux: parameter gradients of this layer
            e._sid = AUX if dec_on_aux else SIDE
            e._wait(P, "dec_dg%d" % l)
            gw = "decoder.rnn.weight_hh_l%d" % l
            grp = [] if (e.group_wgrads and dt == L.BF16) else None      # dW_hh (two pieces) and dW_ih of the layer: one grid
            if Tp > 1:
                e._gemm(P, L.GEMM_TN, dg.p(B), dg.ld, outb.p(0, ocol), outb.ld, e.gp(gw), H, 4 * Hp, H, (Tp - 1) * B, out_f32=1, split_k=-1,
                        rmap=gmap_d, group=grp)
            e._gemm(P, L.GEMM_TN, dg.p(0), dg.ld, self.hn[l].p(), self.hn[l].ld, e.gp(gw), H, 4 * Hp, H, B, out_f32=1, split_k=-1, rmap=gmap_d,
                    group=grp)
            gi = "decoder.rnn.weight_ih_l%d" % l
            # the bias gradient (column sums of dgates) rides in the dW_ih product, which reads all M rows of dgates anyway
            bsum = (None, 0, e.gp("decoder.rnn.bias_ih_l%d" % l), e.gp("decoder.rnn.bias_hh_l%d" % l)) if e.lstm_db_in_gemm else None
            if l == 0:
                if self.z_in_Xt:
                    # dW_ih [4H][E + Z] = dgates^T [emb ; z]: one product over the rows of Xt (z repeated over the steps from column Ep on);
                    # the column blocks [0, Ep) / [Ep, Ep + Z) land at columns [0, E) / [E, E + Z) of the gradient
                    fused_b = e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, self.Xt.p(), self.Xt.ld, e.gp(gi, 0, 0), E + Z, 4 * Hp, self.Ep + Z, M, out_f32=1,
                                      split_k=-1, colsum=bsum, rmap=gmap_d, cmap=(self.Ep, E), group=grp)
                    e._gemm_group(P, grp)
                else:
                    fused_b = e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, self.Xt.p(), self.Xt.ld, e.gp(gi, 0, 0), E + Z, 4 * Hp, E, M, out_f32=1, split_k=-1,
                                      colsum=bsum, rmap=gmap_d, group=grp)
                    e._gemm_group(P, grp)
                    e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, self.zT.p(), self.zT.ld, e.gp(gi, 0, E), E + Z, 4 * Hp, Z, M, out_f32=1, split_k=-1, b_kmod=B,
                            rmap=gmap_d)
                we = e.sh["dec_wih_l0_e"]
                # dX = dgates W_e, then its rows scattered into the embedding gradient (pad row dropped) -- on the SIDE stream, idle while
                # the encoder's backward recurrence runs, next to this stream's weight gradients instead of behind them
                dxt_side = dec_on_aux
                if dxt_side:
                    e._sid = SIDE
                    e._wait(P, "dec_dg%d" % l)
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, we.p(), we.ld, self.dXt.p(), self.dXt.ld, M, E, 4 * Hp, out_f32=1)
                e._call(P, lib.vmmt_scatter_add_rows, self.dXt.p(), self.dXt.ld, self.tgt_in.data_ptr(), PAD,
                        e.gp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E, M, E)
                if spread and dxt_side and e.rows_active():
                    P.append((None, (1, 4), "SUMSQ_ROWS", None, SIDE))
                    self._tgt_rows_normed = True
                if dxt_side:
                    e._sid = AUX
            else:
                xin = self.dec_xdrop[l - 1] if drop else self.dec_out[l - 1]
                fused_b = e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, xin.p(), xin.ld, e.gp(gi), H, 4 * Hp, H, M, out_f32=1, split_k=-1, colsum=bsum,
                                  rmap=gmap_d, group=grp)
                e._gemm_group(P, grp)
            if not fused_b:
                e._call(P, lib.vmmt_colsum, dt, dg.p(), dg.ld, M, 4 * Hp, Hp if Hp != H else 0, H, e.gp("decoder.rnn.bias_ih_l%d" % l),
                        e.gp("decoder.rnn.bias_hh_l%d" % l))
        if rp:
            # reparameterised gradient (H2 switched off): dL/dz = sum_t dgates_t W_z (decoder input, VI_Model1.py:99-100) + the image
            # network's gate path; it joins the KL gradient at mu / sigma, so the q(z|x) networks' backward can only start here,
            # behind the decoder chain
            e._sid = AUX
            e._wait(P, "dec_dg0")
            dg0, wz = self.dec_dgates[0], e.sh["dec_wih_l0_z"]
            e._gemm(P, L.GEMM_NN, dg0.p(), dg0.ld, wz.p(), wz.ld, self.dzrow.p(), self.dzrow.ld, M, Z, 4 * Hp, out_f32=1)
            e._call(P, lib.vmmt_reparam_dz, self.dzrow.p(), self.dzrow.ld, Tp, self.dzt.p(), self.dzt.ld, self.z32.p(),
                    self.gate.data_ptr(), e.pp("inf_net_image.gate_affine_transform.weight"), self.dz.p(), B, Z)
            kl_and_q_backward()
        e._sid = SIDE
        if dec_on_aux:
            e._record(P, "side_first")              # generator + attention products and the target-embedding gradient issued on the side stream

        def finish_first_half():
            e._allreduce(P, "decoder.attn.linear_out.weight", "encoder.rnn.weight_ih_l%d" % (Lyr - 1))
            # gradient norm of everything that is final by now (attention, decoder, inference networks; the generator's went out behind
            # its own product): off the critical path, underneath the encoder chain
            e._sumsq_entry(P, "decoder.attn.linear_out.weight" if gen_norm_early else "generator.0.weight",
                           "encoder.rnn.weight_ih_l%d" % (Lyr - 1), 0, skip_rows=self._tgt_rows_normed)
            e._wait(P, "aux_done")
            if not self._inf_normed:
                e._sumsq_entry(P, "inf_net_image.location.fc2.weight", None, 2)
        if dec_on_aux:
            e._sid = AUX
            e._wait(P, "side_first")
            finish_first_half()
            e._record(P, "aux_end")
            e._sid = SIDE
        elif not d.conditional:
            finish_first_half()
        # ================= encoder LSTM backward (main) + its parameter gradients (side) ==========================
        if d.conditional:   # p(z|x) reads the NON-detached memory (Models.py:889): d context[s,b] += d hbar_p[b] / len_b
            e._sid = MAIN
            e._wait(P, "dhbar_p")
            e._call(P, lib.vmmt_masked_mean_bwd, dt, self.dhbar_p.p(), self.dhbar_p.ld, self.src_len.data_ptr(), self.dctx.p(),
                    self.dctx.ld, B, S, H, 0, 1)
        dh_above = self.dctx
        def enc_param_grads(l, ranges, alternate):
            """dW_hh, db, dW_ih (and for layer 0 the embedding scatter) of encoder layer l from the time steps [lo, hi) of each
            direction; every product accumulates into the arena, so ranges may be issued separately"""
            dg = self.enc_dgates[l]
            wih = e.sh["enc_wih_l%d" % l]
            xin = (self.Xs if l == 0 else (self.enc_xdrop[l - 1] if drop else self.enc_out[l - 1]))
            xcols = E if l == 0 else H
            tog = [e._sid]
            grp = [] if (e.group_wgrads and dt == L.BF16) else None      # dW_hh and dW_ih of both directions: one grid on the side stream
            if grp is not None and l == 0:
                # ... next to the embedding product, which goes out FIRST on the main stream (the grid of the group fills the chip)
                assert all(r == (0, S) for r in ranges)
                e._sid = MAIN
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, wih.p(), wih.ld, self.dXs.p(), self.dXs.ld, MS, E, dirs * 4 * Hdp, out_f32=1)
                e._call(P, lib.vmmt_scatter_add_rows, self.dXs.p(), self.dXs.ld, self.src.data_ptr(), PAD,
                        e.gp("encoder.embeddings.make_embedding.emb_luts.0.weight"), E, MS, E)
                if e.tail_norm_first and e.rows_active() and not e.dp_on():
                    # the source table's flagged rows are final here: their norm right behind the scatter, beside the side stream's grid,
                    # instead of in the step's tail behind the join
                    P.append((None, (0, 3), "SUMSQ_ROWS", None, MAIN))
                    self._src_rows_normed = True
                e._sid = SIDE

            def alt():
                if alternate and grp is None:
                    tog[0] = MAIN if tog[0] == SIDE else SIDE
                    e._sid = tog[0]
            for k, suf in enumerate([""] + (["_reverse"] if d.brnn else [])):
                lo, hi = ranges[k]
                gw = "encoder.rnn.weight_hh_l%d%s" % (l, suf)
                alt()
                if k == 0:      # h_prev[t] = out[t-1]: t in [max(lo, 1), hi)
                    t0 = max(lo, 1)
                    if hi > t0:
                        e._gemm(P, L.GEMM_TN, dg.p(t0 * B, k * 4 * Hdp), dg.ld, self.enc_out[l].p((t0 - 1) * B, k * Hdp), self.enc_out[l].ld,
                                e.gp(gw), Hd, 4 * Hdp, Hd, (hi - t0) * B, out_f32=1, split_k=-1, rmap=gmap_e, group=grp)
                else:           # h_prev[t] = out[t+1]: t in [lo, min(hi, S-1))
                    t1 = min(hi, S - 1)
                    if t1 > lo:
                        e._gemm(P, L.GEMM_TN, dg.p(lo * B, k * 4 * Hdp), dg.ld, self.enc_out[l].p((lo + 1) * B, k * Hdp), self.enc_out[l].ld,
                                e.gp(gw), Hd, 4 * Hdp, Hd, (t1 - lo) * B, out_f32=1, split_k=-1, rmap=gmap_e, group=grp)
                alt()           # (alternating: main = the two dW_hh and the embedding product behind them, side = dW_ih + bias sums)
                bih, bhh = e.gp("encoder.rnn.bias_ih_l%d%s" % (l, suf)), e.gp("encoder.rnn.bias_hh_l%d%s" % (l, suf))
                fused_b = e._gemm(P, L.GEMM_TN, dg.p(lo * B, k * 4 * Hdp), dg.ld, xin.p(lo * B), xin.ld, e.gp("encoder.rnn.weight_ih_l%d%s" % (l, suf)),
                                  xcols, 4 * Hdp, xcols, (hi - lo) * B, out_f32=1, split_k=-1, rmap=gmap_e,
                                  colsum=(None, 0, bih, bhh) if e.lstm_db_in_gemm else None, group=grp)      # bias gradient: see the decoder's
                if not fused_b:
                    e._call(P, lib.vmmt_colsum, dt, dg.p(lo * B, k * 4 * Hdp), dg.ld, (hi - lo) * B, 4 * Hdp, Hdp if Hdp != Hd else 0, Hd, bih, bhh)
            e._gemm_group(P, grp)
            if l == 0 and grp is None:                # embedding gradient: one product over both directions, last on the main stream
                assert all(r == (0, S) for r in ranges)
                e._sid = MAIN
                # (as the GEMM's own atomic epilogue the scatter costs 47 us on top of a 28 us product in isolation; in the step the two
                #  forms measure the same -- 2.07-2.08 ms, tools/ab.py -- the row kernel is kept for its simpler access pattern)
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, wih.p(), wih.ld, self.dXs.p(), self.dXs.ld, MS, E, dirs * 4 * Hdp, out_f32=1)
                e._call(P, lib.vmmt_scatter_add_rows, self.dXs.p(), self.dXs.ld, self.src.data_ptr(), PAD,
                        e.gp("encoder.embeddings.make_embedding.emb_luts.0.weight"), E, MS, E)

        # two or more layers: the TOP encoder layer's backward recurrence needs the top decoder layer's state gradients and d context
        # only, not the lower decoder layers -- it runs on a stream of its own next to them (dec_l1 -> {dec_l0 || enc_l1} -> enc_l0
        # instead of four recurrences in a row).  Two persistent launches share the chip only when their workgroups fit side by side
        # (the scripts' batch of 40: 2 x 64 workgroups; at 256 sentences each launch fills the chip and they run one after the other
        # as before)
        # ... and ONLY then are they put on two streams: two persistent launches that start at the same moment without room for both
        # could each get a part of their workgroups resident and wait for the rest until the in-launch waits time out.
        seq_wgs = lambda hh, nd: min(256, -(-B // 32) * (hh // 16) * nd)
        par_top = bool(e.bwd_layers_parallel and Lyr >= 2 and not d.conditional and not rp and e.use_side_stream and e.use_aux_stream
                       and seq_wgs(Hdp, dirs) + seq_wgs(Hp, 1) <= 256)
        for l in reversed(range(Lyr)):
            on_tgt = par_top and l == Lyr - 1
            e._sid = TGT if on_tgt else MAIN
            if on_tgt:
                e._wait(P, "dec_dg%d" % l)
            elif par_top and l == Lyr - 2:
                e._wait(P, "enc_top_dx")
            dg = self.enc_dgates[l]
            seq = (L.LstmDirBwd * (S * dirs))()
            for step in range(S):
                for k in range(dirs):
                    # backward visits the steps in the reverse of the forward order of that direction
                    t = (S - 1 - step) if k == 0 else step
                    tn = (t + 1) if k == 0 else (t - 1)        # step processed just before (its dgates feed the GEMM)
                    tp = (t - 1) if k == 0 else (t + 1)        # forward predecessor (c_prev)
                    a = seq[step * dirs + k]
                    whhT = e.sh["enc_whhT_l%d_d%d" % (l, k)]
                    if step > 0:
                        a.dgates_next, a.ld_dgn = dg.p(tn * B, k * 4 * Hdp), dg.ld
                    a.w_hh_t, a.ld_wt = whhT.p(), whhT.ld
                    a.dh_above, a.ld_dha = dh_above.p(t * B, k * Hdp), dh_above.ld
                    a.gates, a.ld_gates = self.enc_gates[l].p(t * B, k * 4 * Hdp), self.enc_gates[l].ld
                    a.c_t, a.ld_ct = self.enc_c[l].p(t * B, k * Hdp), self.enc_c[l].ld
                    if 0 <= tp < S:
                        a.c_prev, a.ld_cp = self.enc_c[l].p(tp * B, k * Hdp), self.enc_c[l].ld
                    a.dc_carry, a.ld_dcc = self.enc_dcc[l].p(0, k * Hdp), self.enc_dcc[l].ld
                    a.dgates_out, a.ld_dgo = dg.p(t * B, k * 4 * Hdp), dg.ld
                    a.dh_n, a.ld_dhn = self.dec_dh0[l].p(0, k * Hdp), self.dec_dh0[l].ld
                    a.dc_n, a.ld_dcn = self.dec_dcc[l].p(0, k * Hdp), self.dec_dcc[l].ld
                    a.t = t
                    a.inject = 1 if k == 0 else 2
            # (the encoder's parameter gradients are the step's tail; cutting this recurrence into two launches so that the products of
            #  the first half of its steps run next to the second half was measured: 2.000 against 1.881 ms -- the relaunch has to wait
            #  for the product's workgroups to leave before all of its own are resident)
            e._lstm_seq_bwd(P, seq, dirs, S, self.src_len.data_ptr(), B, Hdp)
            e._record(P, "enc_dg%d" % l)
            wih = e.sh["enc_wih_l%d" % l]
            if l > 0:
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, wih.p(), wih.ld, self.enc_dx[l - 1].p(), self.enc_dx[l - 1].ld, MS, H, dirs * 4 * Hdp)
                if drop:
                    e._call(P, lib.vmmt_mul, dt, self.enc_dx[l - 1].p(), self.enc_dx[l - 1].ld, self.enc_mask[l - 1].p(),
                            self.enc_mask[l - 1].ld, self.enc_dx[l - 1].p(), self.enc_dx[l - 1].ld, MS, H)
                dh_above = self.enc_dx[l - 1]
            if on_tgt:
                e._record(P, "enc_top_dx")
            e._sid = SIDE
            e._wait(P, "enc_dg%d" % l)
            enc_param_grads(l, [(0, S)] * dirs, l == 0)
        if d.conditional:
            e._sid = SIDE
            e._wait(P, "aux_done")
            finish_first_half()
        # join
        e._sid = SIDE
        if spread and self._src_rows_normed:
            # (the encoder's weight gradients are this stream's last products: their norm here, beside the main stream's scatter + row norm)
            e._allreduce(P, "encoder.rnn.weight_ih_l%d" % (Lyr - 1), "inf_net_image.location.fc2.weight")
            e._sumsq_entry(P, "encoder.rnn.weight_ih_l%d" % (Lyr - 1), "inf_net_image.location.fc2.weight", 1, skip_rows=True)
            self._enc_dense_normed = True
        e._record(P, "side_done")
        e._sid = MAIN
        e._wait(P, "side_done")
        # (tail_norm_first: the encoder segment's norm in front of the join with the aux stream -- whose own norms of the first half are the
        #  last thing it runs -- instead of behind it; the update behind this plan needs both.  Not under data parallelism: the collectives'
        #  issue order is part of the protocol)
        norm_first = bool(dec_on_aux and getattr(e, "tail_norm_first", False) and not e.dp_on())
        if dec_on_aux and not norm_first:
            e._wait(P, "aux_end")
        if not self._enc_dense_normed:
            e._allreduce(P, "encoder.rnn.weight_ih_l%d" % (Lyr - 1), "inf_net_image.location.fc2.weight")
            e._sumsq_entry(P, "encoder.rnn.weight_ih_l%d" % (Lyr - 1), "inf_net_image.location.fc2.weight", 1, skip_rows=self._src_rows_normed)
        if norm_first:
            e._wait(P, "aux_end")
        return P

    # ------------------------------------------------------------------------------- conditional-prior variant (8f-1)
