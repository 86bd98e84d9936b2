"""VI_Model1 training-step engine: owns the HBM layout (parameter arena, compute shadows, per-shape workspace)
and drives the libvmmt.so kernels.  PyTorch is used for device memory, streams and (in dp.py) torch.distributed --
no torch operator computes anything on the hot path.

Reference path (all under /root/reference): TrainerMultimodal._gradient_accumulation
(onmt/TrainerMultimodal.py:625-718) -> NMTVIModel.forward (onmt/Models.py:850-1011) ->
NMTVIModel1LossCompute.sharded_compute_loss (onmt/Loss.py:88-132, onmt/VILoss.py:217-513) -> Optim.step
(onmt/Optim.py:78-96).

HBM layout
  * arena: ONE flat fp32 buffer for all master parameters (views carry the reference's state-dict names,
    SURVEY.md Appendix B), one for gradients, two for Adam moments.  Parameters that never receive a gradient
    (inf_net_image.scale.*, hazard H6) sit at the tail, outside the optimiser / all-reduce range.  The order is
    the order in which backward finishes gradients (generator first, embeddings last) so that data-parallel
    buckets can be reduced while backward is still running.
  * shadows: compute copies of the 2-D weights in the storage type T (bf16 or fp32), leading dimension padded
    to 16 bytes, refreshed by vmmt_pack after each optimiser step.
  * workspace: activations saved for backward, per (B, S, T') shape, time-major rows (t*B + b).
"""
import collections
import ctypes as C
import math
from os import environ as _os_env

import torch

from . import _lib as L

PAD = 1  # '<blank>' (onmt/io/DatasetBase.py:7-11)


def _ru(x, m):
    return (x + m - 1) // m * m


class Dims(object):
    def __init__(self, vs, vt, emb=500, hid=500, z=500, img=2048, layers=2, brnn=False, dropout=0.0, conditional=False):
        self.vs, self.vt, self.emb, self.hid, self.z, self.img = vs, vt, emb, hid, z, img
        self.layers, self.brnn, self.dropout = layers, bool(brnn), float(dropout)
        self.conditional = bool(conditional)       # --conditional prior (ModelConstructor.py:435-460)
        self.ht = hid // 2                          # encoder_tgt is always bidirectional (ModelConstructor.py:456-457)
        self.qin = 2 * hid + img if conditional else hid
        assert not conditional or hid % 2 == 0
        self.dirs = 2 if brnn else 1
        assert hid % self.dirs == 0
        self.hd = hid // self.dirs
        assert hid <= 1024, "attention kernel limit (H <= 1024)"
        # COMPUTE layout of the hidden size.  The run scripts train -rnn_size 500 --z_latent_dim 500, 2-layer uni-directional
        # (run_translated_m30k_only.sh:46-57, opts.py:14-16,54,67-69); the MFMA LSTM / attention kernels tile H in 32s and the persistent
        # recurrences serve H in {64, 128, 256, 512}.  So hidden vectors are computed `hp` wide (500 -> 512, gate g of a 4H vector at
        # g * hp) with zeros in the padding: shadows are packed gate block by gate block, pre-activations / h / c / every gradient
        # are exactly zero in padded lanes (sigmoid(0) * tanh(0)), and gradients are stored back through the block map of
        # vmmt_gemm_args.c_row_blk.  The arena, the state dict, checkpoints, Adam and the all-reduce keep the reference shapes.
        # (a bidirectional ENCODER with an odd per-direction size keeps the general kernels.)  The conditional model's encoder_tgt is
        # always bidirectional with hid / 2 units per direction (250 -> 256): its output is laid out [fwd | pad | bwd | pad], 2 * htp
        # wide, and feeds the posterior network's input [h_x : hp | h_y : 2 htp | v : img] (qin_p columns).
        self.pad = (not self.brnn) and hid % 32 != 0 and _os_env.get("VMMT_PAD_HIDDEN", "1") == "1"
        self.hp = _ru(hid, 32) if self.pad else hid
        self.hdp = self.hp // self.dirs
        self.htp = _ru(self.ht, 32) if self.pad else self.ht
        self.qin_p = (self.hp + 2 * self.htp + img) if self.conditional else self.hp
        self.zp = _ru(z, 128)                       # tiled latent size of the fused q(z|x) kernel (Z_valid = z)

    def param_shapes(self):
        """name -> shape, in ARENA order (reverse of backward completion is not needed: order == completion)."""
        d = self
        s = []
        s += [("generator.0.weight", (d.vt, d.hid)), ("generator.0.bias", (d.vt,))]
        s += [("decoder.attn.linear_out.weight", (d.hid, 2 * d.hid)), ("decoder.attn.linear_in.weight", (d.hid, d.hid))]
        for l in reversed(range(d.layers)):
            i = d.emb + d.z if l == 0 else d.hid
            s += [("decoder.rnn.weight_ih_l%d" % l, (4 * d.hid, i)), ("decoder.rnn.weight_hh_l%d" % l, (4 * d.hid, d.hid)),
                  ("decoder.rnn.bias_ih_l%d" % l, (4 * d.hid,)), ("decoder.rnn.bias_hh_l%d" % l, (4 * d.hid,))]
        s += [("decoder.embeddings.make_embedding.emb_luts.0.weight", (d.vt, d.emb))]
        for l in reversed(range(d.layers)):
            i = d.emb if l == 0 else d.hid
            for suf in ([""] + (["_reverse"] if d.brnn else [])):
                s += [("encoder.rnn.weight_ih_l%d%s" % (l, suf), (4 * d.hd, i)),
                      ("encoder.rnn.weight_hh_l%d%s" % (l, suf), (4 * d.hd, d.hd)),
                      ("encoder.rnn.bias_ih_l%d%s" % (l, suf), (4 * d.hd,)),
                      ("encoder.rnn.bias_hh_l%d%s" % (l, suf), (4 * d.hd,))]
        s += [("encoder.embeddings.make_embedding.emb_luts.0.weight", (d.vs, d.emb))]
        if d.conditional:
            for br in ("location", "scale"):       # p(z|x)
                s += [("gen_net_global.%s.fc2.weight" % br, (d.z, d.z)), ("gen_net_global.%s.fc2.bias" % br, (d.z,)),
                      ("gen_net_global.%s.fc1.weight" % br, (d.z, d.hid)), ("gen_net_global.%s.fc1.bias" % br, (d.z,))]
            for l in reversed(range(d.layers)):    # encoder_tgt (shares the decoder's embedding table)
                i = d.emb if l == 0 else d.hid
                for suf in ("", "_reverse"):
                    s += [("encoder_tgt.rnn.weight_ih_l%d%s" % (l, suf), (4 * d.ht, i)),
                          ("encoder_tgt.rnn.weight_hh_l%d%s" % (l, suf), (4 * d.ht, d.ht)),
                          ("encoder_tgt.rnn.bias_ih_l%d%s" % (l, suf), (4 * d.ht,)),
                          ("encoder_tgt.rnn.bias_hh_l%d%s" % (l, suf), (4 * d.ht,))]
        s += [("inf_net_image.location.fc2.weight", (d.img, d.img)), ("inf_net_image.location.fc2.bias", (d.img,)),
              ("inf_net_image.location.fc1.weight", (d.img, d.z)), ("inf_net_image.location.fc1.bias", (d.img,)),
              ("inf_net_image.gate_affine_transform.weight", (1, d.z)), ("inf_net_image.gate_affine_transform.bias", (1,))]
        for br in ("location", "scale"):
            s += [("inf_net_global.%s.fc2.weight" % br, (d.z, d.z)), ("inf_net_global.%s.fc2.bias" % br, (d.z,)),
                  ("inf_net_global.%s.fc1.weight" % br, (d.z, d.qin)), ("inf_net_global.%s.fc1.bias" % br, (d.z,))]
        nograd = [("inf_net_image.scale.fc1.weight", (d.img, d.z)), ("inf_net_image.scale.fc1.bias", (d.img,)),
                  ("inf_net_image.scale.fc2.weight", (d.img, d.img)), ("inf_net_image.scale.fc2.bias", (d.img,))]
        return s, nograd


KPAD = 64          # GEMM reduction slab (elements)
SEG_ALIGN = 512    # arena segments start on multiples of this many elements (8 ranks x 64-element units)


class Buf(object):
    """2-D device buffer [rows][ld].  Rows and (unless `ld` is given) columns are zero-padded to whole 64-element GEMM
    slabs plus one spare slab of rows, so that a GEMM may round its reduction length K up to a multiple of 64 whichever
    way the buffer is traversed (K-contiguous or K-strided, also from a row / column offset): the padding contributes
    exact zeros.  Nothing ever writes the padding."""

    def __init__(self, rows, cols, dtype, device, ld=None, fill=None, storage=None):
        esz = torch.empty((), dtype=dtype).element_size()
        self.ld = ld if ld is not None else _ru(max(cols, 1), KPAD)
        self.rows, self.cols, self.esz = rows, cols, esz
        prow = _ru(max(rows, 1), KPAD) + KPAD
        if storage is not None:
            # a view of storage shared between workspaces (Engine.shared_storage): it holds FINITE leftovers of other shapes
            # instead of zeros; only for buffers whose every reduction partner is zero-padded itself (see Workspace.GT)
            self.t = storage[:prow * self.ld].view(prow, self.ld)
        else:
            self.t = torch.zeros(prow, self.ld, dtype=dtype, device=device)
        if fill is not None:
            self.t[:rows, :cols].fill_(fill)

    @staticmethod
    def elems(rows, cols):
        return (_ru(max(rows, 1), KPAD) + KPAD) * _ru(max(cols, 1), KPAD)

    def p(self, r=0, c=0):
        return self.t.data_ptr() + (r * self.ld + c) * self.esz

    def view(self):
        return self.t[:self.rows, :self.cols]


class Engine(object):
    def __init__(self, dims, dtype="bf16", device="cuda", seed=0, param_init=0.1, batch_global=None):
        self.d = dims
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("variational_mmt_amd needs a GPU (MI355X); there is no CPU path")
        self.lib = L.lib()
        self.row_tables = []
        self.dt = L.BF16 if dtype in ("bf16", torch.bfloat16) else L.F32
        self.T = torch.bfloat16 if self.dt == L.BF16 else torch.float32
        self.tsz = 2 if self.dt == L.BF16 else 4
        self._build_arena(seed, param_init)
        self._build_row_tables()
        self._build_shadows()
        # per-shape workspaces: a bounded LRU (real data has hundreds of (S, T') pairs per epoch).  Shapes are rounded up to
        # `shape_bucket` positions (pad positions are masked everywhere), the largest buffer (G^T) is shared between all
        # workspaces, and the least recently used workspaces are dropped once `ws_budget_bytes` is exceeded.
        self.ws = collections.OrderedDict()
        self.shape_bucket = max(1, int(_os_env.get("VMMT_SHAPE_BUCKET", "2")))
        self.ws_budget_bytes = int(float(_os_env.get("VMMT_WS_BUDGET_GB", "48")) * (1 << 30))
        self.ws_evictions = 0
        self._shared = {}
        self.shadows_dirty = True
        self.step_count = 0          # Adam step counter
        self.seed = seed
        self.rng_counter = 1234567 + seed      # dp.GradSync offsets it by the rank: eps / dropout masks differ between replicas
        # H2: as executed the reference detaches the latent sample (Dists.py:21-26, Models.py:930-933): mu / sigma get gradient only
        # through the KL.  True = the reparameterised estimator the paper describes (d z flows from the decoder input and the
        # image network into q(z|x)); tested against the oracle's reparam_grad=True (tests/test_gpu_step_parity.py)
        self.reparam_grad = False
        self.stats_host = None
        self._sid = 0
        # the side stream carries bulk work that is off the critical path: lowest hardware priority, so that small
        # critical-path kernels on the main stream are dispatched ahead of its queued workgroups
        try:
            pr = torch.cuda.Stream.priority_range()
            lo_pri, hi_pri = max(pr), min(pr)
        except Exception:
            lo_pri, hi_pri = 0, 0
        self.side_stream = self._side_stream_plain = torch.cuda.Stream(device=self.dev, priority=lo_pri)
        # third stream: the small, latency-bound backward of the image / q(z|x) networks, independent of the text path
        self.aux_stream = torch.cuda.Stream(device=self.dev, priority=lo_pri)
        # measured (tools/sched_ab.py, fixed prior): 2.53 ms/step with that work on the side stream vs 2.59 ms on a stream of
        # its own; the conditional model keeps it (encoder_tgt's 2 x B-step recurrences would otherwise block the side stream)
        self.use_aux_stream = _os_env.get("VMMT_AUX_STREAM", "1") == "1"
        # fourth stream (conditional model only): encoder_tgt's backward recurrence, 2 x B dependent steps that would otherwise hold
        # up everything queued behind them on the aux stream
        self.tgt_stream = torch.cuda.Stream(device=self.dev, priority=lo_pri)
        # workgroups of the BACKGROUND half of Adam (side stream, underneath the next step's encoder recurrence).  Unthrottled it takes
        # the memory system for ~200 us and the latency-bound persistent LSTM kernel next to it runs at half speed; one workgroup per
        # CU still finishes before the decoder-side weights are needed.  tools/ab.py, ms per step: 4096 wgs 1.958-2.007 | 384: 1.957 |
        # 288: 1.972 | 256: 1.921-1.951 | 224: 1.908 | 192: 1.987 | 128: 2.051 | 64: 2.324 (a faster two-chunk kernel at 256: 1.98)
        self.bg_adam_blocks = int(_os_env.get("VMMT_BG_ADAM_BLOCKS", "256"))
        # cap of the weight-gradient products' split-K.  On an idle chip 4, 8 and 16 splits cost the same (tools/gemm_split.py), in the step
        # the extra workgroups and atomics get in the way of everything that runs next to them (tools/ab.py, ms per step by cap: 64:
        # 1.854, 8: 1.834-1.843, 6: 1.798, 5: 1.782, 4: 1.792-1.812, 3: 1.815, 2: 1.848, 1: 2.003)
        self.max_split_k = int(_os_env.get("VMMT_MAX_SPLIT_K", "4"))
        self.cond_aux_early = _os_env.get("VMMT_COND_AUX_EARLY", "1") == "1"
        self.cond_emb_fg = _os_env.get("VMMT_COND_EMB_FG", "1") == "1"
        self.aux_early = _os_env.get("VMMT_AUX_EARLY", "1") == "1"
        self.aux_kl_first = _os_env.get("VMMT_AUX_KL_FIRST", "1") == "1"
        self.gen_db_in_gemm = _os_env.get("VMMT_GEN_DB_IN_GEMM", "1") == "1"
        self.lstm_db_in_gemm = _os_env.get("VMMT_LSTM_DB_IN_GEMM", "1") == "1"
        self.dec_grads_on_aux = _os_env.get("VMMT_DEC_GRADS_ON_AUX", "1") == "1"
        self.bwd_main_first = _os_env.get("VMMT_BWD_MAIN_FIRST", "1") == "1"      # issue order of the backward plan (see _plan_backward)
        self.bwd_layers_parallel = _os_env.get("VMMT_BWD_LAYERS_PARALLEL", "1") == "1"   # >= 2 layers: top encoder layer next to the lower decoder layers
        # (a high-priority stream for the critical path was measured and is slightly SLOWER than the default stream:
        #  tools/sched_ab.py, 3.249 vs 3.226 ms/step)
        self.compute_stream = torch.cuda.Stream(device=self.dev, priority=hi_pri)
        self.use_side_stream = True
        self._masked_streams = []
        import os as _os
        self.q_parallel = _os.environ.get("VMMT_QPAR", "1") == "1"    # q(z|x): scale branch on the side stream next to the location branch
        self.trace = None            # list -> _run appends (name, timing event) at every main-stream phase change
        self.global_events = {}      # events that outlive a plan run (optimizer <-> next forward)
        self.split_optim = True      # run the decoder-side half of Adam + shadow refresh on the side stream
        self._sumsq = torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=self.dev)   # slot totals | tickets | partials (vmmt.h)
        self._sumsq_by_plan = False
        self.fused_qnet = _os_env.get("VMMT_FUSED_QNET", "1") == "1"
        self.qnet_split = _os_env.get("VMMT_QNET_SPLIT", "1") == "1"     # location / scale networks in separate workgroups (csrc/qnet.hip)
        self.gen_fused = _os_env.get("VMMT_GEN_FUSED", "1") == "1"       # csrc/generator_fused.hip where it applies (bf16, H = 512 / 256)
        # decode.py: a decoded position as ONE hipGraph, replayed -- built, bit-identical, and measured SLOWER than issuing its ~15 launches
        # one by one (tools/decode_bench.py, ms per 24 positions, graph / plain: beam 5 x 30 sentences 3.87 / 3.62, arg-max x 256 3.68 / 2.92:
        # a position is bound by the GPU's dependent-kernel turnaround, not by the host, and a replay does not overlap the next one's launch)
        self.decode_graphs = _os_env.get("VMMT_DECODE_GRAPHS", "0") == "1"
        self.persistent_lstm = _os_env.get("VMMT_PERSISTENT_LSTM", "1") == "1"     # plans are built per workspace: set before the first forward
        self.seq_syncs = []
        self.dp = None               # dp.GradSync when torch.distributed runs with > 1 rank
        self._works = []

    def set_side_cu_mask(self, mask_words):
        """restrict the side stream to the CUs set in `mask_words` (list of 32-bit words, bit i = CU i); None restores the
        unrestricted low-priority stream.  Keeps CUs free for the main stream's latency-critical kernels."""
        if mask_words is None:
            self.side_stream = self._side_stream_plain
            return
        arr = (C.c_uint32 * len(mask_words))(*[int(w) & 0xFFFFFFFF for w in mask_words])
        out = C.c_void_p()
        L.check(self.lib.vmmt_stream_create_masked(arr, len(mask_words), 0, C.byref(out)), "vmmt_stream_create_masked")
        self._masked_streams.append(out.value)
        self.side_stream = torch.cuda.ExternalStream(out.value, device=self.dev)

    # ------------------------------------------------------------------------------------------------ arena
    def _build_arena(self, seed, param_init):
        wg, ng = self.d.param_shapes()
        self.names_grad = [n for n, _ in wg]
        self.names_nograd = [n for n, _ in ng]
        off = 0
        self.offsets = {}
        self.first_enc_name = "encoder.rnn.weight_ih_l%d" % (self.d.layers - 1)     # arena: [generator|attn|decoder|dec emb][encoder|enc emb|inference nets]
        # the four data-parallel SEGMENTS of the arena, in the order backward completes them (_plan_backward issues one collective per
        # segment): [generator][attention + decoder + target embeddings][encoder + source embeddings (+ conditional networks)]
        # [inference networks].  A segment starts on a multiple of SEG_ALIGN elements, so that it splits into 1 / 2 / 4 / 8 equal
        # rank shards of whole 64-element units (reduce-scatter + sharded Adam + all-gather: optim_step); the padding holds zeros
        seg_starts = ("generator.0.weight", "decoder.attn.linear_out.weight", self.first_enc_name, "inf_net_image.location.fc2.weight")
        self.seg_bounds = []
        for n, shp in wg + ng:
            if n in seg_starts or n == ng[0][0]:
                off = _ru(off, SEG_ALIGN)
                self.seg_bounds.append(off)
            if n == ng[0][0]:
                self.n_opt = off                       # optimiser / all-reduce range = [0, n_opt)
            self.offsets[n] = (off, shp)
            off += _ru(int(math.prod(shp)), 64)
        self.n_total = off
        assert len(self.seg_bounds) == 5 and self.seg_bounds[-1] == self.n_opt
        self.segments = list(zip(self.seg_bounds[:-1], self.seg_bounds[1:]))
        dev = self.dev
        self.flat_p = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.n_opt, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(self.n_opt, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(self.n_opt, dtype=torch.float32, device=dev)
        self.params, self.grads = {}, {}
        g = torch.Generator(device="cpu").manual_seed(seed)       # (explicit devices: a driver may have made CUDA the default tensor type)
        for n, (o, shp) in self.offsets.items():
            k = int(math.prod(shp))
            self.params[n] = self.flat_p[o:o + k].view(*shp)
            if o < self.n_opt:
                self.grads[n] = self.flat_g[o:o + k].view(*shp)
            if param_init:                             # ModelConstructor.py:598-603 (H7: every parameter)
                self.params[n].copy_(((torch.rand(*shp, generator=g, device="cpu") * 2 - 1) * param_init))

    # ------------------------------------------------------------------------------------------------ embedding tables by row
    def _build_row_tables(self):
        """Row-wise gradient bookkeeping for the two embedding tables (csrc/optim.hip: vmmt_rows_mark / _zero / vmmt_sumsq_rows /
        vmmt_adam_rows_step).  The tables are 54 % of the optimised parameters and a step's gradient lives in the <= S B + T' B rows
        the batch looked up (17 % of 30 000 at the benchmark shape): with one flag per row, the gradient is cleared, normed and read
        for those rows only -- 12 of the 36 B per element and step the dense path moves (4 zeroing + 4 norm + 28 Adam), and the
        zero-fill and the norm stop touching 200 MB each.  Every row is still UPDATED at every step (the moments of a row without
        gradient decay, its parameter follows them): bit-identical to dense Adam (tests/test_gpu_row_adam.py).  Off under data
        parallelism (the flagged set would have to be the union over the ranks) and for the conditional model (two streams flag
        rows of the shared target table).
        OPT-IN (VMMT_ROW_ADAM=1): measured on MI355X it does not pay at these sizes -- 1.787 against 1.746 ms per step at BASELINE
        config 2, 2.704 against 2.728 at the run scripts' shape, 1.867 against 1.863 through the trainer: the dense streams run at
        6.7 TB/s, half of them underneath the next step's encoder, while the row kernels add six small launches to the step's head
        and tail.  (The LAZY variant -- rows updated only when used, missed zero-gradient steps replayed -- saves 0.7 GB per step and
        was 1.5 % faster on the benchmark's recurring batches, but with Zipf-distributed ids the replays (sqrt + division per element
        and missed step, in front of the embedding lookup) cost more than the traffic: 2.11 against 1.92 ms through the trainer.  Not
        kept: DESIGN.md section 6.)"""
        names = ("encoder.embeddings.make_embedding.emb_luts.0.weight", "decoder.embeddings.make_embedding.emb_luts.0.weight")
        self.row_adam = _os_env.get("VMMT_ROW_ADAM", "0") == "1" and not self.d.conditional
        self.row_tables = []
        if self.d.conditional:
            return
        for n in names:          # (the flag arrays are always there -- 240 KB -- so that the switch can be set after construction)
            off, (R, Cc) = self.offsets[n]
            if Cc % 4 or off % 4:
                self.row_adam, self.row_tables = False, []
                return
            self.row_tables.append(dict(name=n, off=off, R=R, C=Cc, end=off + R * Cc,
                                        flags=torch.zeros(R, dtype=torch.int32, device=self.dev),
                                        rowsq=torch.zeros(R, dtype=torch.float32, device=self.dev)))

    def rows_active(self):
        return bool(self.row_tables) and self.row_adam and not (self.dp is not None and self.dp.world > 1)

    def _row_mark_entries(self, plan, table_index, ids_ptr, n_ids):
        """plan entries (training forward, off the critical path): flag the batch's rows of an embedding table and clear their
        gradient rows, which the backward plan's scatter-add accumulates into"""
        if not self.rows_active():
            return
        t = self.row_tables[table_index]
        self._call(plan, self.lib.vmmt_rows_mark, ids_ptr, n_ids, t["flags"].data_ptr(), t["R"])
        self._call(plan, self.lib.vmmt_rows_zero, self.flat_g.data_ptr() + 4 * t["off"], t["R"], t["C"], t["flags"].data_ptr())

    def pp(self, name, r=0, c=0):
        o, shp = self.offsets[name]
        ld = shp[1] if len(shp) > 1 else 0
        return self.flat_p.data_ptr() + (o + r * ld + c) * 4

    def gp(self, name, r=0, c=0):
        o, shp = self.offsets[name]
        ld = shp[1] if len(shp) > 1 else 0
        return self.flat_g.data_ptr() + (o + r * ld + c) * 4

    def load_state_dict(self, sd):
        for n, t in sd.items():
            if n in self.params:
                self.params[n].copy_(t.to(torch.float32))
        self.shadows_dirty = True

    def state_dict(self):
        sd = {n: v.detach().clone() for n, v in self.params.items()}
        if self.d.conditional:      # encoder_tgt shares the decoder's table; the reference's state dict lists it under both names
            sd["encoder_tgt.embeddings.make_embedding.emb_luts.0.weight"] = sd["decoder.embeddings.make_embedding.emb_luts.0.weight"]
        return sd

    # ------------------------------------------------------------------------------------------------ shadows
    def _build_shadows(self):
        d, T, dev = self.d, self.T, self.dev
        self.sh = {}
        self.pack_calls = []

        def shadow(key, rows, cols, src, c0=0, ncols=None, transpose=False, dtype=None, src2=None, row_off=0, col_off=0, gate=None):
            """one compute copy (or one piece of it).  (row_off, col_off): position of the piece in the shadow as stored, i.e. AFTER the
            transpose.  gate = (h, hp), h != hp: the source's 4h rows (a bias: its 4h entries) are nn.LSTM's gate blocks i, f, g, o; block
            g lands at g * hp of the shadow (Dims.hp): one pack descriptor per block"""
            dt = dtype if dtype is not None else T
            code = L.F32 if dt == torch.float32 else L.BF16
            _, shp = self.offsets[src]
            two_d = len(shp) > 1
            ld_src = shp[1] if two_d else shp[0]
            R = shp[0] if two_d else 1
            Cc = ncols if ncols is not None else (shp[1] if two_d else shp[0])
            if key not in self.sh:
                self.sh[key] = Buf(rows, cols, dt, dev)
            b = self.sh[key]
            blocks = [(0, 0)] if (gate is None or gate[0] == gate[1]) else [(g * gate[0], g * gate[1]) for g in range(4)]
            for s0, d0 in blocks:
                if len(blocks) > 1:
                    if two_d:
                        R = gate[0]
                    else:
                        Cc = gate[0]
                if two_d:       # gate blocks are row blocks of the source: row blocks of the shadow, column blocks of a transposed one
                    sp = self.pp(src, s0, c0)
                    s2 = self.pp(src2, s0, c0) if src2 else None
                    dst = b.p(row_off + d0, col_off) if not transpose else b.p(row_off, col_off + d0)
                else:           # a bias vector [4h] packed as one row
                    sp = self.pp(src, 0, s0)
                    s2 = self.pp(src2, 0, s0) if src2 else None
                    dst = b.p(row_off, col_off + d0)
                self.pack_calls.append((code, sp, s2, ld_src, dst, b.ld, R, Cc, 1 if transpose else 0))

        ge, gd = (d.hd, d.hdp), (d.hid, d.hp)       # gate blocks of the encoder's directions / of the decoder as stored -> as computed
        for l in range(d.layers):
            i = d.emb if l == 0 else d.hid
            for k, suf in enumerate([""] + (["_reverse"] if d.brnn else [])):
                # concatenated over directions: rows k*4Hd ..
                shadow("enc_wih_l%d" % l, d.dirs * 4 * d.hdp, i, "encoder.rnn.weight_ih_l%d%s" % (l, suf), row_off=k * 4 * d.hdp, gate=ge)
                shadow("enc_b_l%d" % l, 1, d.dirs * 4 * d.hdp, "encoder.rnn.bias_ih_l%d%s" % (l, suf), dtype=torch.float32,
                       src2="encoder.rnn.bias_hh_l%d%s" % (l, suf), col_off=k * 4 * d.hdp, gate=ge)
                shadow("enc_whh_l%d_d%d" % (l, k), 4 * d.hdp, d.hd, "encoder.rnn.weight_hh_l%d%s" % (l, suf), gate=ge)
                shadow("enc_whhT_l%d_d%d" % (l, k), d.hd, 4 * d.hdp, "encoder.rnn.weight_hh_l%d%s" % (l, suf), transpose=True, gate=ge)
        for l in range(d.layers):
            if l == 0:
                shadow("dec_wih_l0_e", 4 * d.hp, d.emb, "decoder.rnn.weight_ih_l0", c0=0, ncols=d.emb, gate=gd)
                shadow("dec_wih_l0_z", 4 * d.hp, d.z, "decoder.rnn.weight_ih_l0", c0=d.emb, ncols=d.z, gate=gd)
            else:
                shadow("dec_wih_l%d" % l, 4 * d.hp, d.hid, "decoder.rnn.weight_ih_l%d" % l, gate=gd)
            shadow("dec_b_l%d" % l, 1, 4 * d.hp, "decoder.rnn.bias_ih_l%d" % l, dtype=torch.float32,
                   src2="decoder.rnn.bias_hh_l%d" % l, gate=gd)
            shadow("dec_whh_l%d" % l, 4 * d.hp, d.hid, "decoder.rnn.weight_hh_l%d" % l, gate=gd)
            shadow("dec_whhT_l%d" % l, d.hid, 4 * d.hp, "decoder.rnn.weight_hh_l%d" % l, transpose=True, gate=gd)
        shadow("wa", d.hid, d.hid, "decoder.attn.linear_in.weight")
        # W_out [H][2H] multiplies [c ; r] (GlobalAttention.py:187); the two halves of that buffer start at 0 and hp
        if d.hp == d.hid:
            shadow("wo", d.hid, 2 * d.hid, "decoder.attn.linear_out.weight")
        else:
            shadow("wo", d.hid, 2 * d.hp, "decoder.attn.linear_out.weight", c0=0, ncols=d.hid)
            shadow("wo", d.hid, 2 * d.hp, "decoder.attn.linear_out.weight", c0=d.hid, ncols=d.hid, col_off=d.hp)
        for br in ("location", "scale"):
            if d.conditional and d.pad:
                # W1 [Z][h_x (H) | h_y (2 ht) | v] against the padded input [h_x : hp | h_y fwd : htp | h_y bwd : htp | v]
                for c0, nc, co in ((0, d.hid, 0), (d.hid, d.ht, d.hp), (d.hid + d.ht, d.ht, d.hp + d.htp), (2 * d.hid, d.img, d.hp + 2 * d.htp)):
                    shadow("q_%s_w1" % br, d.z, d.qin_p, "inf_net_global.%s.fc1.weight" % br, c0=c0, ncols=nc, col_off=co)
            else:
                shadow("q_%s_w1" % br, d.z, d.qin, "inf_net_global.%s.fc1.weight" % br)
            shadow("q_%s_w2" % br, d.z, d.zp, "inf_net_global.%s.fc2.weight" % br)     # (read d.zp columns wide by the fused q(z|x) kernel)
        if d.conditional:
            for br in ("location", "scale"):
                shadow("p_%s_w1" % br, d.z, d.hid, "gen_net_global.%s.fc1.weight" % br)
                shadow("p_%s_w2" % br, d.z, d.z, "gen_net_global.%s.fc2.weight" % br)
            gt = (d.ht, d.htp)
            for l in range(d.layers):
                for k, suf in enumerate(("", "_reverse")):
                    nm = "encoder_tgt.rnn.weight_ih_l%d%s" % (l, suf)
                    if l == 0 or d.htp == d.ht:
                        shadow("enct_wih_l%d" % l, 2 * 4 * d.htp, d.emb if l == 0 else d.hid, nm, row_off=k * 4 * d.htp, gate=gt)
                    else:       # the layer below delivers [fwd : htp | bwd : htp]
                        shadow("enct_wih_l%d" % l, 2 * 4 * d.htp, 2 * d.htp, nm, row_off=k * 4 * d.htp, gate=gt, c0=0, ncols=d.ht)
                        shadow("enct_wih_l%d" % l, 2 * 4 * d.htp, 2 * d.htp, nm, row_off=k * 4 * d.htp, gate=gt, c0=d.ht, ncols=d.ht, col_off=d.htp)
                    shadow("enct_b_l%d" % l, 1, 2 * 4 * d.htp, "encoder_tgt.rnn.bias_ih_l%d%s" % (l, suf), dtype=torch.float32,
                           src2="encoder_tgt.rnn.bias_hh_l%d%s" % (l, suf), col_off=k * 4 * d.htp, gate=gt)
                    shadow("enct_whh_l%d_d%d" % (l, k), 4 * d.htp, d.ht, "encoder_tgt.rnn.weight_hh_l%d%s" % (l, suf), gate=gt)
                    shadow("enct_whhT_l%d_d%d" % (l, k), d.ht, 4 * d.htp, "encoder_tgt.rnn.weight_hh_l%d%s" % (l, suf), transpose=True, gate=gt)
        shadow("iv_w1", d.img, d.z, "inf_net_image.location.fc1.weight")
        shadow("iv_w2", d.img, d.img, "inf_net_image.location.fc2.weight")
        shadow("wg", d.vt, d.hid, "generator.0.weight")

    def _fused_shadows(self):
        """[(lo, hi, shadow pointer)] in arena order: optimised 2-D weights of at least 2 M elements whose bf16 shadow has unpadded rows
        and no second source -- the shadow then has the parameter's flat layout and vmmt_adam_step writes it (optim_step)"""
        if not hasattr(self, "_fused_sh"):
            self._fused_sh = []
            base = self.flat_p.data_ptr()
            for code, sp, s2, lds, dst, ldd, R, Cc, tr in self.pack_calls:
                o = (sp - base) // 4
                whole = any(off == o and len(shp) == 2 and shp[0] * shp[1] == R * Cc for off, shp in self.offsets.values())
                if code == L.BF16 and not tr and s2 is None and ldd == Cc and lds == Cc and whole and R * Cc >= (1 << 21) and \
                        o + R * Cc <= self.n_opt and dst % 8 == 0:
                    self._fused_sh.append((o, o + R * Cc, dst))
            self._fused_sh.sort()
        return self._fused_sh

    def _pack_tables(self):
        """descriptor tables for vmmt_pack_multi.  Parts 0 / 1: every shadow of [encoder + inference networks] / [generator + attention +
        decoder] (after load_state_dict / a replica broadcast); parts 2 / 3: the same without the shadows the optimiser step writes
        itself (_fused_shadows): what optim_step refreshes"""
        if not hasattr(self, "_pack_tab"):
            self._pack_tab = []
            enc_lo = self.offsets[self.first_enc_name][0]
            base = self.flat_p.data_ptr()
            fused = set(d for _, _, d in self._fused_shadows())
            for part in (0, 1, 2, 3):
                sel = [c for c in self.pack_calls if ((c[1] - base) // 4 >= enc_lo) == (part % 2 == 0) and (part < 2 or c[4] not in fused)]
                arr = (L.PackDesc * max(1, len(sel)))()
                start = 0
                for k, (code, sp, s2, lds, dst, ldd, R, Cc, tr) in enumerate(sel):
                    ch = ((R + 63) // 64) * ((Cc + 31) // 32) if tr else (R * Cc + 2047) // 2048      # vmmt.h: vmmt_pack_multi
                    arr[k] = L.PackDesc(sp, s2, dst, lds, ldd, R, Cc, tr, code, start, ch)
                    start += ch
                host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                self._pack_tab.append((host.to(self.dev), len(sel), start))
        return self._pack_tab

    def _pack_part(self, part, stream):
        tab, n, chunks = self._pack_tables()[part]
        if n:
            L.check(self.lib.vmmt_pack_multi(tab.data_ptr(), n, chunks, stream), "vmmt_pack_multi")

    def refresh_shadows(self, stream):
        """fp32 master -> compute shadows (after load_state_dict / first use; optim_step refreshes them itself)."""
        if not self.shadows_dirty:
            return
        self._pack_part(0, stream)
        self._pack_part(1, stream)
        self.shadows_dirty = False

    # ------------------------------------------------------------------------------------------------ helpers
    def _gemm(self, plan, layout, A, lda, B, ldb, Cp, ldc, M, N, K, addend=None, ld_add=0, add_rows=0, add_is_T=0,
              act=L.ACT_NONE, out_f32=0, accumulate=0, alpha=1.0, a_kmod=0, b_kmod=0, scatter_ids=None, tile=0, split_k=0,
              b_batch_rows=0, b_batch_stride=0, colsum=None, rmap=None, cmap=None):
        """plan entry: one vmmt_gemm.  colsum = (w, w_stride, out[, out2]): the column sums of the K-strided A operand from the same
        pass (weighted by w, or plain with w = None), where the library offers them; returns whether they were attached"""
        if split_k == -1:
            # weight-gradient heuristic: enough workgroups to fill 256 CUs, >= 256 reduction steps each, at most max_split_k splits
            tiles = ((M + 63) // 64) * ((N + 63) // 64)
            split_k = max(1, min(K // 256, (1024 + tiles - 1) // tiles, int(self.max_split_k)))
            if split_k == 1:
                accumulate = 1          # gradients always ACCUMULATE into the arena (zeroed at the start of a step)
        if a_kmod == 0 and b_kmod == 0:
            K = _ru(K, KPAD)            # operands are Bufs: zero-padded to whole slabs (see Buf)
        a = L.GemmArgs(self.dt, layout, A, lda, B, ldb, Cp, ldc, M, N, K, a_kmod, b_kmod, addend, ld_add, add_rows,
                       add_is_T, act, out_f32, accumulate, alpha, scatter_ids, PAD, tile, split_k, b_batch_rows, b_batch_stride,
                       None, 0, None, None)
        # rmap / cmap = (blk, valid): output rows / columns computed in padded blocks, stored densely (vmmt_gemm_args.c_row_blk)
        if rmap is not None and rmap[0] != rmap[1]:
            a.c_row_blk, a.c_row_valid = rmap
        if cmap is not None and cmap[0] != cmap[1]:
            a.c_col_blk, a.c_col_valid = cmap
        attached = False
        if colsum is not None:
            a.colsum_w, a.colsum_w_stride, a.colsum_out = colsum[:3]
            a.colsum_out2 = colsum[3] if len(colsum) > 3 else None
            attached = bool(self.lib.vmmt_gemm_colsum_applies(C.byref(a)))
            if not attached:
                a.colsum_w, a.colsum_w_stride, a.colsum_out, a.colsum_out2 = None, 0, None, None
        plan.append((self.lib.vmmt_gemm, (C.byref(a),), "gemm", a, self._sid))
        return attached

    def _call(self, plan, fn, *args):
        # plan entries are positional ctypes calls: at least the COUNT is checked against the declared signature when the plan is built
        # (the stream is appended at run time), so that a kernel that gained an argument fails here and not as a shifted pointer
        assert fn.argtypes is not None and len(args) + 1 == len(fn.argtypes), (fn.__name__, len(args) + 1, len(fn.argtypes))
        plan.append((fn, args, fn.__name__, None, self._sid))

    # pointer fields of the step descriptors that walk the batch: (pointer, leading dimension, element size; None = the storage type T)
    _SEQ_F = (("h_prev", "ld_hprev", None), ("c_prev", "ld_cprev", 4), ("gx", "ld_gx", 4), ("gx2", "ld_gx2", 4), ("gates", "ld_gates", None),
              ("c_out", "ld_c", 4), ("h_out", "ld_h", None), ("h_n", "ld_hn", None), ("c_n", "ld_cn", 4))
    _SEQ_B = (("dgates_next", "ld_dgn", None), ("dh_above", "ld_dha", None), ("gates", "ld_gates", None), ("c_t", "ld_ct", 4), ("c_prev", "ld_cp", 4),
              ("dc_carry", "ld_dcc", 4), ("dgates_out", "ld_dgo", None), ("dh_n", "ld_dhn", 4), ("dc_n", "ld_dcn", 4), ("dh0_out", "ld_dh0", 4))

    def _seq_row_chunks(self, arr, fields, ndir, B, H):
        """A persistent recurrence needs all of its workgroups resident at once: (B / 32 row groups) x (H / 16 unit slices) x directions
        <= 256.  Sentences are independent in a recurrence, so a batch that does not fit is cut into ROW chunks, one persistent launch
        each, one after the other (BASELINE config 5: H = 1024 -> 64 slices -> 128 sentences per launch; the per-step kernels it
        replaces re-read their W_hh slice from L2 at every step: 35 us per backward step against 8).  -> [(descriptors, row offset,
        rows)], or None when the persistent kernel does not serve this size at all."""
        if H not in (64, 128, 256, 512, 1024):
            return None
        groups = 256 // ((H // 16) * ndir)
        if groups < 1:
            return None
        rows = 32 * groups
        if B <= rows:
            return [(arr, 0, B)]
        out = []
        n = len(arr)
        for r0 in range(0, B, rows):
            chunk = (type(arr[0]) * n)()
            C.memmove(chunk, arr, C.sizeof(arr))
            for a in chunk:
                for ptr, ld, esz in fields:
                    p0 = getattr(a, ptr)
                    if p0:
                        setattr(a, ptr, p0 + r0 * getattr(a, ld) * (esz if esz is not None else self.tsz))
            out.append((chunk, r0, min(rows, B - r0)))
        return out

    def _lstm_seq_fwd(self, plan, arr, ndir, nsteps, lens_ptr, B, H):
        """plan entry: a whole forward recurrence (nsteps x ndir step descriptors in `arr`).  persistent_lstm: ONE launch of the
        persistent kernel (W_hh resident in LDS, in-launch hand-off of h_t: csrc/lstm_seq.hip), which falls back by itself to the
        per-step kernels where it does not apply; otherwise the per-step kernels issued from one host call."""
        chunks = self._seq_row_chunks(arr, self._SEQ_F, ndir, B, H) if (self.persistent_lstm and self.dt == L.BF16) else None
        if chunks is not None and len(chunks) > 1:
            for sub, r0, rows in chunks:
                self._lstm_seq_fwd(plan, sub, ndir, nsteps, (lens_ptr + 8 * r0) if lens_ptr else None, rows, H)
            return
        if self.persistent_lstm:
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
            sync = torch.zeros(self.lib.vmmt_lstm_seq_sync_words(), dtype=torch.int32, device=self.dev)
            xchg = torch.zeros(max(16, self.lib.vmmt_lstm_seq_xchg_bytes(ndir, B, H)), dtype=torch.uint8, device=self.dev)
            plan.append((self.lib.vmmt_lstm_seq_fwd, (self.dt, ndir, nsteps, arr, dev.data_ptr(), lens_ptr, B, H, sync.data_ptr(), xchg.data_ptr()),
                         "vmmt_lstm_seq_fwd", (arr, dev, sync, xchg), self._sid))
            self.seq_syncs.append(sync)
        else:
            plan.append((self.lib.vmmt_lstm_chain_fwd, (self.dt, ndir, nsteps, arr, lens_ptr, B, H), "vmmt_lstm_chain_fwd", arr, self._sid))

    def _lstm_seq_bwd(self, plan, arr, ndir, nsteps, lens_ptr, B, H, with_dh0=0):
        """plan entry: a whole backward recurrence (the mode-0 steps; with_dh0: `arr` ends with one mode-1 step, the gradient of
        the initial hidden state), see _lstm_seq_fwd"""
        chunks = self._seq_row_chunks(arr, self._SEQ_B, ndir, B, H) if (self.persistent_lstm and self.dt == L.BF16) else None
        if chunks is not None and len(chunks) > 1:
            for sub, r0, rows in chunks:
                self._lstm_seq_bwd(plan, sub, ndir, nsteps, (lens_ptr + 8 * r0) if lens_ptr else None, rows, H, with_dh0=with_dh0)
            return
        if self.persistent_lstm:
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
            sync = torch.zeros(self.lib.vmmt_lstm_seq_sync_words(), dtype=torch.int32, device=self.dev)
            xchg = torch.zeros(max(16, self.lib.vmmt_lstm_seq_xchg_bytes_bwd(ndir, B, H)), dtype=torch.uint8, device=self.dev)
            plan.append((self.lib.vmmt_lstm_seq_bwd, (self.dt, ndir, nsteps, arr, dev.data_ptr(), lens_ptr, B, H, with_dh0, sync.data_ptr(),
                                                      xchg.data_ptr()), "vmmt_lstm_seq_bwd", (arr, dev, sync, xchg), self._sid))
            self.seq_syncs.append(sync)
        else:
            plan.append((self.lib.vmmt_lstm_chain_bwd, (self.dt, ndir, nsteps, arr, lens_ptr, B, H, 0), "vmmt_lstm_chain_bwd", arr, self._sid))
            if with_dh0:
                last = C.cast(C.byref(arr, nsteps * ndir * C.sizeof(L.LstmDirBwd)), C.POINTER(L.LstmDirBwd))
                plan.append((self.lib.vmmt_lstm_step_bwd, (self.dt, ndir, last, lens_ptr, B, H, 1), "vmmt_lstm_step_bwd", arr, self._sid))

    def lstm_seq_errors(self):
        """error words of the persistent recurrence launches so far (0 = every in-launch wait completed); synchronises"""
        torch.cuda.synchronize(self.dev)
        words = self.lib.vmmt_lstm_seq_sync_words()
        return [int(s[2].item()) for s in self.seq_syncs]            # [launch epoch, finish count, error word]

    def check_async_errors(self):
        """raise if an in-launch wait of a persistent recurrence kernel ever ran into its 2-second bound (a workgroup of the row group
        was not resident: another process on the GPU, a CU mask): the results of that step are then wrong.  Synchronises; the trainer
        mirror calls it at the end of every epoch and before a checkpoint is written, bench.py after its timed region."""
        bad = [i for i, x in enumerate(self.lstm_seq_errors()) if x != 0]
        if bad:
            raise RuntimeError("persistent LSTM launch(es) %s reported a hand-off timeout (error words %s): results invalid; "
                               "rerun with VMMT_PERSISTENT_LSTM=0" % (bad, [self.lstm_seq_errors()[i] for i in bad]))

    def _zero(self, plan, tensors):
        """plan entry: clear all `tensors` (contiguous device tensors) with ONE vmmt_zero_multi launch"""
        arr = (L.ZeroDesc * len(tensors))()
        start = 0
        for k, t in enumerate(tensors):
            nbytes = t.numel() * t.element_size()
            assert t.is_contiguous() and t.data_ptr() % 16 == 0 and nbytes % 4 == 0
            arr[k] = L.ZeroDesc(t.data_ptr(), nbytes, start)
            start += (nbytes + 16383) // 16384
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        plan.append((self.lib.vmmt_zero_multi, (tab.data_ptr(), len(tensors), start), "vmmt_zero_multi", (tab, tensors), self._sid))

    # -- two-stream plans: entries carry a stream id (0 = main = torch's current stream, 1 = side stream); EV_RECORD /
    #    EV_WAIT entries fork and join them.  Work that is off the critical path of the step (weight-gradient GEMMs, bias
    #    sums, the image / q(z|x) networks' backward) runs on the side stream underneath the latency-bound LSTM steps.
    def _record(self, plan, ev):
        plan.append((None, ev, "EV_RECORD", None, self._sid))

    def _wait(self, plan, ev):
        plan.append((None, ev, "EV_WAIT", None, self._sid))

    def _allreduce(self, plan, first_name, end_name):
        """plan entry: sum the arena range [offset(first_name), offset(end_name)) over the data-parallel ranks, issued on
        the entry's stream right behind the kernels that produced it (no-op for a single process)."""
        lo = self.offsets[first_name][0]
        hi = self.offsets[end_name][0] if end_name is not None else self.n_opt
        plan.append((None, (lo, min(hi, self.n_opt)), "ALLREDUCE", None, self._sid))

    def _sumsq_entry(self, plan, first_name, end_name, slot):
        """plan entry: ||g||^2 of an arena range into slot `slot` of the step's norm scratch (behind that range's all-reduce);
        every range has a slot of its own and Adam adds the slots in index order: the norm is bit-reproducible"""
        lo = self.offsets[first_name][0]
        hi = self.offsets[end_name][0] if end_name is not None else self.n_opt
        hi = min(hi, self.n_opt)
        if self.rows_active():
            # an embedding table inside the range: its flagged rows go to a slot of their own (3 + table index), the dense kernel
            # takes what lies in front of it (slot) and behind it (slot + 5: only the conditional model has parameters there)
            for k, t in enumerate(self.row_tables):
                if lo <= t["off"] and t["end"] <= hi:
                    plan.append((None, (k, 3 + k), "SUMSQ_ROWS", None, self._sid))
                    if t["off"] > lo:
                        plan.append((None, (lo, t["off"], slot), "SUMSQ", None, self._sid))
                    if hi - t["end"] >= SEG_ALIGN:      # (less: only the segment's alignment padding follows)
                        plan.append((None, (t["end"], hi, slot + 5), "SUMSQ", None, self._sid))
                    return
        plan.append((None, (lo, hi, slot), "SUMSQ", None, self._sid))

    def finish_allreduce(self):
        """make the current stream wait for every outstanding gradient all-reduce (call before optim_step)"""
        for w in self._works:
            w.wait()
        self._works = []

    def _run(self, plan, events=None):
        main = torch.cuda.current_stream(self.dev)
        side = self.side_stream if self.use_side_stream else main
        aux = self.aux_stream if (self.use_side_stream and self.use_aux_stream) else side
        tgt = self.tgt_stream if aux is self.aux_stream else aux
        ts = (main, side, aux, tgt)
        hs = (main.cuda_stream, side.cuda_stream, aux.cuda_stream, tgt.cuda_stream)
        trace, last = self.trace, None
        for entry in plan:
            fn, args, name, _keep, sid = entry
            if trace is not None and sid == 0 and name != last:     # tools/phase_times.py: timing events at phase changes
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(main)
                trace.append((name, ev))
                last = name
            self._exec(entry, ts, hs, events, side is main)

    def _exec(self, entry, ts, hs, events, single_stream):
        fn, args, name, _keep, sid = entry
        if fn is None:
            if name == "SUMSQ":
                if self.dp is not None and self.dp.world > 1 and self.dp.sharded:
                    return          # sharded optimiser: every rank takes the norm of ITS shards in optim_step
                lo, hi, slot = args
                if self._works:
                    with torch.cuda.stream(ts[sid]):
                        for w in self._works:
                            w.wait()
                    self._works = []
                L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * lo, hi - lo, self._sumsq.data_ptr(), slot, hs[sid]), "vmmt_sumsq")
                self._sumsq_by_plan = True
                return
            if name == "SUMSQ_ROWS":
                k, slot = args
                t = self.row_tables[k]
                L.check(self.lib.vmmt_sumsq_rows(self.flat_g.data_ptr() + 4 * t["off"], t["R"], t["C"], t["flags"].data_ptr(), t["rowsq"].data_ptr(),
                                                 self._sumsq.data_ptr(), slot, hs[sid]), "vmmt_sumsq_rows")
                return
            if name == "KL_ALLREDUCE":
                if self.dp is not None and self.dp.world > 1:
                    ws = self._cur_ws
                    with torch.cuda.stream(ts[sid]):
                        ws.kl_global.copy_(ws.stats[L.STAT_KL_SUM:L.STAT_KL_SUM + 1])
                        self.dp.dist.all_reduce(ws.kl_global, async_op=True).wait()
                return
            if name == "ALLREDUCE":
                if self.dp is not None and self.dp.world > 1:
                    lo, hi = args
                    with torch.cuda.stream(ts[sid]):
                        if self.dp.sharded:     # each rank receives the sum of ITS 1/world of the segment (in place)
                            self._works.append(self.dp.reduce_scatter(self.flat_g, lo, hi))
                        else:
                            self._works.append(self.dp.dist.all_reduce(self.flat_g[lo:hi], async_op=True))
                return
            if single_stream:
                return
            if name == "EV_RECORD":
                ev = events.get(args)
                if ev is None:
                    ev = events[args] = torch.cuda.Event()
                ev.record(ts[sid])
            else:
                ev = events.get(args) if args in events else self.global_events.get(args)
                if ev is not None:
                    ts[sid].wait_event(ev)
        else:
            rc = fn(*args, hs[sid])
            if rc != 0:
                L.check(rc, name)

    # ------------------------------------------------------------------------------------------------ workspace
    def bucket_shape(self, S, Tp):
        g = self.shape_bucket
        return min(_ru(S, g), max(S, 64)), _ru(Tp, g)

    def shared_storage(self, name, elems, dtype):
        """one device allocation per name, shared by every workspace and grown to the largest request.  Growing it invalidates
        the pointers baked into the cached launch plans, so every cached workspace is dropped then (rare: a new largest shape)."""
        cur = self._shared.get(name)
        if cur is None or cur.numel() < elems or cur.dtype != dtype:
            if cur is not None:
                self.drop_workspaces()
                self._shared[name] = cur = None
            self._shared[name] = cur = torch.zeros(elems, dtype=dtype, device=self.dev)
        return cur

    def drop_workspaces(self, keep_last=0):
        """evict cached training workspaces (oldest first), keeping the `keep_last` most recently used"""
        keys = [k for k, v in self.ws.items() if isinstance(v, Workspace)]
        victims = keys[:max(0, len(keys) - keep_last)]
        if victims:
            torch.cuda.synchronize(self.dev)      # their buffers may still be in use on the side streams
            for k in victims:
                del self.ws[k]
                self.ws_evictions += 1

    def workspace_bytes(self):
        return sum(v.nbytes for v in self.ws.values() if isinstance(v, Workspace))

    def workspace(self, B, S, Tp):
        """the workspace (buffers + launch plans) serving B sentences, S source positions, T' decoder steps: shape
        (B, bucket(S), bucket(T'))"""
        Sb, Tb = self.bucket_shape(S, Tp)
        key = (B, Sb, Tb)
        ws = self.ws.get(key)
        if ws is not None:
            self.ws.move_to_end(key)
            return ws
        before = torch.cuda.memory_allocated(self.dev)
        ws = Workspace(self, B, Sb, Tb)
        ws.nbytes = max(0, torch.cuda.memory_allocated(self.dev) - before)
        self.ws[key] = ws
        while self.workspace_bytes() > self.ws_budget_bytes and sum(isinstance(v, Workspace) for v in self.ws.values()) > 1:
            oldest = next(k for k, v in self.ws.items() if isinstance(v, Workspace))
            torch.cuda.synchronize(self.dev)
            del self.ws[oldest]
            self.ws_evictions += 1
        return ws


class Workspace(object):
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
        self.Xt = nb(M, E)
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
        else:
            self.GT = Buf(V, M, T, dev, storage=eng.shared_storage("GT", Buf.elems(V, M), T))
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
        # ---- side stream, underneath the encoder: zero the gradient arena (every gradient writer of the backward plan
        #      runs on the side stream), target embeddings and the time-parallel part of the decoder input projection
        e._sid = SIDE
        e._wait(P, "fwd_begin")
        e._record(P, "side_fwd")
        if training:
            # the generator weight gradient (first in the arena, a third of it) is WRITTEN by its one GEMM, not accumulated
            # ... together with the small accumulators of the backward plan (off the critical path instead of in front of
            # their users): one launch
            if e.rows_active():        # the tables' gradient rows are cleared row by row (vmmt_rows_zero)
                g_ranges, lo = [], e.offsets["generator.0.bias"][0]
                for t in sorted(e.row_tables, key=lambda t: t["off"]):
                    g_ranges.append(e.flat_g[lo:t["off"]])
                    lo = t["end"]
                g_ranges.append(e.flat_g[lo:])
                g_ranges = [r for r in g_ranges if r.numel()]
            else:
                g_ranges = [e.flat_g[e.offsets["generator.0.bias"][0]:]]
            e._zero(P, g_ranges + [e._sumsq[:L.SUMSQ_SLOTS], self.dh1v32.t, self.dzt.t] +
                    ([] if self.gen_fused else [self.dO32.t]) +       # (fused generator: dO is stored, not accumulated)
                    [b.t for l in range(Lyr) for b in (self.dec_dcc[l], self.enc_dcc[l])])
        self._mask_entries = getattr(self, "_mask_entries", {})
        if drop:
            # output dropout mask (VI_Model1.py:132): only needed after the decoder -> generated in the background
            self._mask_entries["dec_out"] = (len(P), self.out_mask)
            e._call(P, lib.vmmt_dropout_mask, dt, self.out_mask.p(), self.out_mask.rows * self.out_mask.ld, d.dropout, 0)
            e._record(P, "out_mask")
        e._call(P, lib.vmmt_gather_rows, dt, e.pp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                self.tgt_in.data_ptr(), self.Xt.p(), self.Xt.ld, M, E)
        we = e.sh["dec_wih_l0_e"]
        e._gemm(P, L.GEMM_NT, self.Xt.p(), self.Xt.ld, we.p(), we.ld, self.dec_gx[0].p(), self.dec_gx[0].ld, M, 4 * Hp, E, out_f32=1)
        if training:
            # row-wise gradient bookkeeping of the embedding tables: flag this batch's rows and clear their gradient rows -- for the
            # backward plan's scatter-adds and the optimiser step; in front of `dec_gx`, which every later stream waits for
            e._row_mark_entries(P, 1, self.tgt_in.data_ptr(), M)
            e._row_mark_entries(P, 0, self.src.data_ptr(), MS)
        e._record(P, "dec_gx")
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
        e._call(P, lib.vmmt_gather_rows, dt, e.pp("encoder.embeddings.make_embedding.emb_luts.0.weight"), E,
                self.src.data_ptr(), self.Xs.p(), self.Xs.ld, MS, E)
        # a3 encoder
        x, xcols = self.Xs, E
        for l in range(Lyr):
            wih, bsum = e.sh["enc_wih_l%d" % l], e.sh["enc_b_l%d" % l]
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
        e._call(P, lib.vmmt_gate_fwd, dt, self.z32.p(), e.pp("inf_net_image.gate_affine_transform.weight"),
                e.pp("inf_net_image.gate_affine_transform.bias"), self.gate.data_ptr(), self.zt.p(), self.zt.ld, B, Z)
        w1, w2 = e.sh["iv_w1"], e.sh["iv_w2"]
        e._gemm(P, L.GEMM_NT, self.zt.p(), self.zt.ld, w1.p(), w1.ld, self.h1v.p(), self.h1v.ld, B, D, Z,
                addend=e.pp("inf_net_image.location.fc1.bias"), ld_add=D, add_rows=1, act=L.ACT_RELU)
        e._gemm(P, L.GEMM_NT, self.h1v.p(), self.h1v.ld, w2.p(), w2.ld, self.mu_v.p(), self.mu_v.ld, B, D, D,
                addend=e.pp("inf_net_image.location.fc2.bias"), ld_add=D, add_rows=1, out_f32=1)
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
        if drop:
            e._wait(P, "out_mask")
            e._call(P, lib.vmmt_mul, dt, self.AH.p(), self.AH.ld, self.out_mask.p(), self.out_mask.ld, self.O.p(), self.O.ld, M, H)
        e._wait(P, "img_fwd")            # join: the loss plans read mu_v
        return P

    def _plan_loss(self, training):
        """forward part of NMTVIModel1LossCompute._compute_loss (VILoss.py:217-513): statistics only."""
        e, d, lib = self.e, self.e.d, self.e.lib
        P = []
        wg = e.sh["wg"]
        O = self.O if (training and d.dropout > 0) else self.AH      # eval: nn.Dropout is the identity
        if training:
            self._loss_patch = None
        if training and self.gen_fused:
            # statistics AND dO = dL/dO in one sweep of Wg + the kernel that folds its vocabulary slices (1 / normalization is patched into
            # the latter by loss_backward: argument 10); the softmax weights P and the per-slice scaled copies O'_s of O they leave
            # behind feed the dWg GEMM of the backward plan
            assert O.ld == _ru(d.hid, KPAD)
            Kp = _ru(d.hid, KPAD)
            e._call(P, lib.vmmt_gen_fwd_dO, e.dt, wg.p(), wg.ld, wg.t.shape[0], e.pp("generator.0.bias"), O.p(), O.ld, self.y.data_ptr(), self.M, d.vt,
                    Kp, self.gen_ws.data_ptr(), self.tgt_logit.data_ptr(), self.gen_P.data_ptr(), self.gen_ldp)
            self._loss_patch = (len(P), 10)
            e._call(P, lib.vmmt_gen_fwd_combine, e.dt, wg.p(), wg.ld, O.p(), O.ld, self.y.data_ptr(), self.M, d.vt, Kp, PAD, 0.0,
                    self.gen_ws.data_ptr(), self.tgt_logit.data_ptr(), self.lse.data_ptr(), self.tok_nll.data_ptr(),
                    self.gen_y32.data_ptr(), self.dO32.p(), self.dO32.ld, self.stats.data_ptr(), self.gen_cs.data_ptr(),
                    self.gen_Os.data_ptr(), self.gen_Os.shape[2], self.gen_Os.shape[1] * self.gen_Os.shape[2])
            return P
        e._call(P, lib.vmmt_gen_loss_fwd, e.dt, wg.p(), wg.ld, e.pp("generator.0.bias"), O.p(), O.ld, self.y.data_ptr(),
                self.M, d.vt, _ru(d.hid, KPAD), PAD, self.part_max.data_ptr(), self.part_sum.data_ptr(), None,
                self.tgt_logit.data_ptr(), self.lse.data_ptr(), self.tok_nll.data_ptr(), self.stats.data_ptr())
        return P

    # ---------------------------------------------------------------------------------------------- backward plan
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
        # (the bias gradient = row sums of G^T comes out of the same kernel; the gradient arena was zeroed on the side stream in
        #  front of `dec_gx`, which the main stream has waited for)
        fuse_db = _os_env.get("VMMT_FUSE_DB", "1") == "1"
        # entries that carry run-time scalars (1 / normalization, KL weights): patched per step by backward_plan(), so that
        # token normalisation (a different value every batch) does not rebuild the plan
        # (walking the vocabulary in 2-6 chunks, so that a chunk of G^T is consumed by dO / dWg while it is still in the Infinity
        #  Cache, was measured with tools/ab.py: 2.227-2.97 ms against 2.213 ms in one pass -- not kept)
        self._patch = {}
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
        # fixed prior: the KL / q(z|x) backward needs mu, sigma and the KL sum only -- it goes first, gated by the sample (and by the side
        # stream's gradient zeroing: dec_gx is recorded behind it), the image term follows when the forward's image network is through
        kl_first = bool(e.aux_early and e.aux_kl_first and not d.conditional and not rp)
        def aux_chain():
            e._sid = AUX
            if kl_first:
                e._wait(P, "z_ready")
                e._wait(P, "dec_gx")
                kl_and_q_backward()
            # the image term and the KL / q(z|x) backward depend on the forward only (mu_v, mu / sigma, the KL sum), not on the generator
            # loss: gated by the forward's image network (behind the step's gradient zeroing on the same stream) they start while the
            # decoder's forward is still running -- the host is a step ahead of the GPU, so the launches are already queued
            early_cond = bool(cond_first and e.aux_early and e.cond_aux_early)
            if early_cond:
                # conditional model: the chain d h_y -> encoder_tgt's backward recurrence is the critical path of the whole backward and
                # depends on the forward only (KL of q against p(z|x)): it starts behind the sample, 0.5 ms before the loss is through
                e._wait(P, "z_ready")
                e._wait(P, "dec_gx")
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
            elif not rp:
                kl_and_q_backward()

        def main_head():
            # main: dropout + tanh backward, linear_out, attention
            e._sid = MAIN
            e._call(P, lib.vmmt_act_bwd, dt, L.ACT_TANH, self.dO32.p(), self.dO32.ld, 1, self.AH.p(), self.AH.ld,
                    self.out_mask.p() if drop else None, self.out_mask.ld if drop else 0, self.dPre.p(), self.dPre.ld, M, H)
            e._record(P, "dPre")
            wo, wa = e.sh["wo"], e.sh["wa"]
            e._gemm(P, L.GEMM_NN, self.dPre.p(), self.dPre.ld, wo.p(), wo.ld, self.dcat.p(), self.dcat.ld, M, 2 * Hp, H)
            ctx = self.enc_out[Lyr - 1]
            e._call(P, lib.vmmt_attn_bwd, dt, self.dcat.p(), self.dcat.ld, self.probs.data_ptr(), self.Q.p(), self.Q.ld, ctx.p(), ctx.ld,
                    self.src_len.data_ptr(), self.dQ.p(), self.dQ.ld, self.dctx.p(), self.dctx.ld, Tp, B, S, Hp)
            e._record(P, "dQ")
            e._gemm(P, L.GEMM_NN, self.dQ.p(), self.dQ.ld, wa.p(), wa.ld, self.dR.p(), self.dR.ld, M, H, H,
                    addend=self.dcat.p(0, Hp), ld_add=self.dcat.ld, add_rows=-1, add_is_T=1)

        # issue order = the order of this list.  With the fused generator dO exists when the plan starts, so the main stream's first
        # kernels go out first instead of behind the ~20 small launches of the aux chain (tools/ab.py: 2.009 against 2.037 ms, equal
        # in a second run); the conditional model keeps the aux chain first, it IS the critical path there (4.42 against 4.45 ms)
        main_first = bool(self.gen_fused and not d.conditional and e.bwd_main_first)
        if main_first:
            main_head()
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
            Og = self.O if (training_dropout and d.dropout > 0) else self.AH
            self._patch["gen"] = (len(P), 10)
            # (bias gradient + one-hot term at the END of the side stream or of the aux stream instead: 1.988 / 1.967 against 1.931-1.934 ms)
            e._call(P, lib.vmmt_gen_dW_finish, dt, self.gen_P.data_ptr(), self.gen_ldp, self.gen_cs.data_ptr(), Og.p(), Og.ld,
                    self.gen_y32.data_ptr(), M, V, Kp, inv_norm, e.gp("generator.0.weight"), H, e.gp("generator.0.bias"), 1 if fused_db else 0)
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
        if main_first:
            aux_chain()
        else:
            main_head()
        e._sid = SIDE
        e._wait(P, "dPre")
        e._gemm(P, L.GEMM_TN, self.dPre.p(), self.dPre.ld, self.cat.p(), self.cat.ld, e.gp("decoder.attn.linear_out.weight"), 2 * H,
                H, 2 * Hp, M, out_f32=1, split_k=-1, cmap=(Hp, H))
        e._wait(P, "dQ")
        e._gemm(P, L.GEMM_TN, self.dQ.p(), self.dQ.ld, self.cat.p(0, Hp), self.cat.ld, e.gp("decoder.attn.linear_in.weight"), H,
                H, H, M, out_f32=1, split_k=-1)
        # the decoder's parameter gradients and the first-half norm go to the AUX stream, which is idle once its own chain (image /
        # q(z|x) networks) is through: behind the generator's products on the side stream they reached into the step's tail
        dec_on_aux = bool(e.dec_grads_on_aux and not rp and not d.conditional and e.use_aux_stream)
        if dec_on_aux:
            e._record(P, "side_first")              # generator + attention products issued on the side stream
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
            if Tp > 1:
                e._gemm(P, L.GEMM_TN, dg.p(B), dg.ld, outb.p(0, ocol), outb.ld, e.gp(gw), H, 4 * Hp, H, (Tp - 1) * B, out_f32=1, split_k=-1,
                        rmap=gmap_d)
            e._gemm(P, L.GEMM_TN, dg.p(0), dg.ld, self.hn[l].p(), self.hn[l].ld, e.gp(gw), H, 4 * Hp, H, B, out_f32=1, split_k=-1, rmap=gmap_d)
            gi = "decoder.rnn.weight_ih_l%d" % l
            # the bias gradient (column sums of dgates) rides in the dW_ih product, which reads all M rows of dgates anyway
            bsum = (None, 0, e.gp("decoder.rnn.bias_ih_l%d" % l), e.gp("decoder.rnn.bias_hh_l%d" % l)) if e.lstm_db_in_gemm else None
            if l == 0:
                fused_b = e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, self.Xt.p(), self.Xt.ld, e.gp(gi, 0, 0), E + Z, 4 * Hp, E, M, out_f32=1, split_k=-1,
                                  colsum=bsum, rmap=gmap_d)
                e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, self.zT.p(), self.zT.ld, e.gp(gi, 0, E), E + Z, 4 * Hp, Z, M, out_f32=1, split_k=-1, b_kmod=B,
                        rmap=gmap_d)
                we = e.sh["dec_wih_l0_e"]
                # dX = dgates W_e, then its rows scattered into the embedding gradient (pad row dropped)
                e._gemm(P, L.GEMM_NN, dg.p(), dg.ld, we.p(), we.ld, self.dXt.p(), self.dXt.ld, M, E, 4 * Hp, out_f32=1)
                e._call(P, lib.vmmt_scatter_add_rows, self.dXt.p(), self.dXt.ld, self.tgt_in.data_ptr(), PAD,
                        e.gp("decoder.embeddings.make_embedding.emb_luts.0.weight"), E, M, E)
            else:
                xin = self.dec_xdrop[l - 1] if drop else self.dec_out[l - 1]
                fused_b = e._gemm(P, L.GEMM_TN, dg.p(), dg.ld, xin.p(), xin.ld, e.gp(gi), H, 4 * Hp, H, M, out_f32=1, split_k=-1, colsum=bsum,
                                  rmap=gmap_d)
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

        def finish_first_half():
            e._allreduce(P, "decoder.attn.linear_out.weight", "encoder.rnn.weight_ih_l%d" % (Lyr - 1))
            # gradient norm of everything that is final by now (generator, attention, decoder, inference networks): off the
            # critical path, underneath the encoder chain
            e._sumsq_entry(P, "generator.0.weight", "encoder.rnn.weight_ih_l%d" % (Lyr - 1), 0)
            e._wait(P, "aux_done")
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

            def alt():
                if alternate:
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
                                e.gp(gw), Hd, 4 * Hdp, Hd, (hi - t0) * B, out_f32=1, split_k=-1, rmap=gmap_e)
                else:           # h_prev[t] = out[t+1]: t in [lo, min(hi, S-1))
                    t1 = min(hi, S - 1)
                    if t1 > lo:
                        e._gemm(P, L.GEMM_TN, dg.p(lo * B, k * 4 * Hdp), dg.ld, self.enc_out[l].p((lo + 1) * B, k * Hdp), self.enc_out[l].ld,
                                e.gp(gw), Hd, 4 * Hdp, Hd, (t1 - lo) * B, out_f32=1, split_k=-1, rmap=gmap_e)
                alt()           # (alternating: main = the two dW_hh and the embedding product behind them, side = dW_ih + bias sums)
                bih, bhh = e.gp("encoder.rnn.bias_ih_l%d%s" % (l, suf)), e.gp("encoder.rnn.bias_hh_l%d%s" % (l, suf))
                fused_b = e._gemm(P, L.GEMM_TN, dg.p(lo * B, k * 4 * Hdp), dg.ld, xin.p(lo * B), xin.ld, e.gp("encoder.rnn.weight_ih_l%d%s" % (l, suf)),
                                  xcols, 4 * Hdp, xcols, (hi - lo) * B, out_f32=1, split_k=-1, rmap=gmap_e,
                                  colsum=(None, 0, bih, bhh) if e.lstm_db_in_gemm else None)      # bias gradient: see the decoder's
                if not fused_b:
                    e._call(P, lib.vmmt_colsum, dt, dg.p(lo * B, k * 4 * Hdp), dg.ld, (hi - lo) * B, 4 * Hdp, Hdp if Hdp != Hd else 0, Hd, bih, bhh)
            if l == 0:                                # embedding gradient: one product over both directions, last on the main stream
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
        e._record(P, "side_done")
        e._sid = MAIN
        e._wait(P, "side_done")
        if dec_on_aux:
            e._wait(P, "aux_end")
        e._allreduce(P, "encoder.rnn.weight_ih_l%d" % (Lyr - 1), "inf_net_image.location.fc2.weight")
        e._sumsq_entry(P, "encoder.rnn.weight_ih_l%d" % (Lyr - 1), "inf_net_image.location.fc2.weight", 1)
        return P

    # ------------------------------------------------------------------------------- conditional-prior variant (8f-1)
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


    def kl_sum_ptr(self):
        """the KL sum the backward weighs against the free-bits margin: this rank's statistic, or under data parallelism its
        all-reduced copy (`kl_global`, filled by the KL_ALLREDUCE entry of the backward plan)"""
        if self.e.dp is not None and self.e.dp.world > 1:
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

    def backward_plan(self, inv_norm, batch_global, kl_mult, use_freebits, margin, drop):
        key = (bool(drop), bool(self.e.reparam_grad))      # the only STRUCTURAL inputs; every scalar is patched below
        if self._bwd_key != key:
            self.plan_bwd = self._plan_backward(inv_norm, batch_global, kl_mult, use_freebits, margin, drop)
            self._bwd_key = key
        P = self.plan_bwd
        fn, _, name, keep, sid = P[self._latent_bwd_index]
        P[self._latent_bwd_index] = (fn, self._latent_bwd_args(batch_global, kl_mult, use_freebits, margin, inv_norm), name, keep, sid)
        for ii, pos in self._patch.values():
            fn, args, name, keep, sid = P[ii]
            P[ii] = (fn, args[:pos] + (float(inv_norm),) + args[pos + 1:], name, keep, sid)
        return P

    def ones_col(self):
        if not hasattr(self, "_ones"):
            self._ones = Buf(self.M, 1, self.e.T, self.e.dev, fill=1.0)
        return self._ones


# ======================================================================================================= step API
def _engine_methods():
    def set_image_table(self, table):
        """`table`: fp32 [N, D] image-feature array (numpy or tensor); kept resident in HBM
        (reference: host numpy + per-step fancy-index + H2D copy, TrainerMultimodal.py:632-639)."""
        t = torch.as_tensor(table)
        self.img_table = t.to(device=self.dev, dtype=torch.float32).contiguous()
        assert self.img_table.shape[1] == self.d.img

    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def forward(self, src, src_len, tgt, img_indices, training=True, eps=None, masks=None, table=None, tgt_len=None):
        """NMTVIModel.forward (Models.py:850-1011).  src [S,B] int64, src_len [B], tgt [T,B] (incl. <s>, </s>),
        img_indices [B] rows of the resident image table.  Returns the Workspace holding every activation."""
        S, B = int(src.shape[0]), int(src.shape[1])
        Tp = int(tgt.shape[0]) - 1
        if S > 64:
            raise RuntimeError("source length %d > 64 not supported by the attention kernel" % S)
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
        L.check(self.lib.vmmt_prepare_batch(src_d.data_ptr(), tgt_d.data_ptr(), len_d.data_ptr(), idx_d.data_ptr(), S, Tp + 1, B,
                                            ws.S, ws.Tp + 1, PAD,
                                            ws.src.data_ptr(), ws.tgt_in.data_ptr(), ws.y.data_ptr(), ws.src_len.data_ptr(),
                                            ws.img_idx.data_ptr(), ws.stats.data_ptr(), ws.eps.p() if gen_eps else None,
                                            B * d.z if gen_eps else 0, self.rng_counter, st), "vmmt_prepare_batch")
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
        if ws._loss_patch is not None:          # fused generator: the statistics pass writes dO = dL/dO scaled by 1 / normalization
            ii, pos = ws._loss_patch
            fn, args, name, keep, sid = ws.plan_loss_train[ii]
            ws.plan_loss_train[ii] = (fn, args[:pos] + (float(1.0 / norm),) + args[pos + 1:], name, keep, sid)
        self._run(ws.plan_loss_train, ws.events)
        plan = ws.backward_plan(1.0 / norm, bg, kl_mult, use_freebits, margin, bool(ws.training))
        self._cur_ws = ws
        self._run(plan, ws.events)
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
        main = torch.cuda.current_stream(self.dev)
        st = main.cuda_stream
        if self.dp is not None and self.dp.world > 1 and self.dp.sharded:
            return self._optim_step_sharded(lr, max_grad_norm, beta1, beta2, eps, grad_scale)
        if max_grad_norm and not self._sumsq_by_plan:      # the backward plan normally accumulates the norm segment by segment
            # (a dense pass is right with the row bookkeeping too: the rows it has not flagged hold zeros)
            self._sumsq[:L.SUMSQ_SLOTS].zero_()
            L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr(), self.n_opt, self._sumsq.data_ptr(), 0, st), "vmmt_sumsq")
        self._sumsq_by_plan = False
        self.step_count += 1
        split = self.offsets[self.first_enc_name][0]
        emb_fg = bool(self.d.conditional and self.cond_emb_fg)
        if emb_fg:
            # conditional model: the shared target embedding table (last item of the background half, no compute shadow) is updated in the
            # FOREGROUND: encoder_tgt's forward recurrence, the step's first long chain, gathers from it right at the start of the step
            split = self.offsets["decoder.embeddings.make_embedding.emb_luts.0.weight"][0]

        def adam_range(lo, hi, stream, shadow=None):
            blocks = int(self.bg_adam_blocks) if stream != st else 0
            if hi > lo:
                L.check(self.lib.vmmt_adam_step(self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
                                                self.flat_m.data_ptr() + 4 * lo, self.flat_v.data_ptr() + 4 * lo, hi - lo, lr, beta1, beta2,
                                                eps, self.step_count, float(max_grad_norm or 0.0), self._sumsq.data_ptr(), grad_scale,
                                                blocks, shadow, stream), "vmmt_adam_step")

        rows = self.rows_active()

        def rows_step(t, stream):
            o = 4 * t["off"]
            L.check(self.lib.vmmt_adam_rows_step(self.flat_p.data_ptr() + o, self.flat_g.data_ptr() + o, self.flat_m.data_ptr() + o,
                                                 self.flat_v.data_ptr() + o, t["R"], t["C"], t["flags"].data_ptr(), lr, beta1, beta2, eps,
                                                 self.step_count, float(max_grad_norm or 0.0), self._sumsq.data_ptr(), grad_scale, stream),
                    "vmmt_adam_rows_step")

        def adam(lo, hi, stream):
            # the big unpadded bf16 shadows (generator weight, image network fc2) are written by the update itself: their range is
            # a launch of its own with the shadow attached, and the shadow refresh behind it skips them (_pack_tables); the embedding
            # tables are updated by the row-wise kernel (gradient read for the batch's rows only) when the row bookkeeping is on
            pieces = [(s_lo, s_hi, ("shadow", ptr)) for s_lo, s_hi, ptr in self._fused_shadows() if lo <= s_lo and s_hi <= hi]
            if rows:
                pieces += [(t["off"], t["end"], ("rows", t)) for t in self.row_tables if lo <= t["off"] and t["end"] <= hi]
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

            def background():
                side.wait_event(ev)
                adam(0, split, side.cuda_stream)
                self._pack_part(3, side.cuda_stream)
                self.global_events.setdefault("opt_side_done", torch.cuda.Event()).record(side)
            background()
        else:
            adam(0, self.n_opt, st)
            self._pack_part(2, st)
            self._pack_part(3, st)
        self.shadows_dirty = False

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
        self.finish_allreduce()                           # the reduce-scatters of the backward plan
        self._sumsq_by_plan = False
        self.step_count += 1
        segs = self.segments
        own = [dp.shard(lo, hi) for lo, hi in segs]
        if max_grad_norm:
            self._sumsq[:L.SUMSQ_SLOTS].zero_()
            for i, (a, b) in enumerate(own):
                if b > a:
                    L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * a, b - a, self._sumsq.data_ptr(), i, st), "vmmt_sumsq")
            tot = dp.all_gather_rows(self._sumsq[:L.SUMSQ_SLOTS])           # [world][SLOTS]
            self._sumsq[:L.SUMSQ_SLOTS].zero_()
            self._sumsq[0:1].copy_(tot.sum(dim=1).sum(dim=0, keepdim=True))  # fixed order: slots of a rank, then the ranks

        def adam(a, b, stream):
            if b > a:
                L.check(self.lib.vmmt_adam_step(self.flat_p.data_ptr() + 4 * a, self.flat_g.data_ptr() + 4 * a, self.flat_m.data_ptr() + 4 * a,
                                                self.flat_v.data_ptr() + 4 * a, b - a, lr, beta1, beta2, eps, self.step_count,
                                                float(max_grad_norm or 0.0), self._sumsq.data_ptr(), grad_scale, 0, None, stream),
                        "vmmt_adam_step")
        fg, bg = (2, 3), (0, 1)                           # foreground: encoder + inference networks; background: generator, decoder
        for i in fg:
            adam(own[i][0], own[i][1], st)
        for i in fg:
            dp.all_gather(self.flat_p, *segs[i]).wait()
        self._pack_part(0, st)
        ev = self.global_events.setdefault("adam_main_done", torch.cuda.Event())
        ev.record(main)
        side = self.side_stream if (self.use_side_stream and self.split_optim) else main
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for i in bg:
                adam(own[i][0], own[i][1], side.cuda_stream)
            for i in bg:
                dp.all_gather(self.flat_p, *segs[i]).wait()
            self._pack_part(1, side.cuda_stream)
        if side is not main:
            self.global_events.setdefault("opt_side_done", torch.cuda.Event()).record(side)
        self.shadows_dirty = False

    for k, v in list(locals().items()):
        if callable(v):
            setattr(Engine, k, v)


_engine_methods()
