"""The engine object: parameter arena, compute shadows, plan execution, workspace cache.  The step API (forward / loss /
backward / optimiser) lives in step.py, the per-shape buffers and launch plans in workspace.py / backward.py / conditional.py."""
import collections
import ctypes as C
import math
from os import environ as _os_env

import torch

from .. import _lib as L
from .layout import Buf, KPAD, PAD, SEG_ALIGN, Dims, _ru  # noqa: F401
from .step import StepAPI
from .workspace import Workspace


def _behind_background(name):
    """attribute that the side-stream half of an optimiser step writes (master parameters, Adam moments): reading it from outside a step
    issues a held-back half first (Engine.bg_after_head) and orders the current stream behind the side stream's half WHETHER HELD BACK OR
    NOT (`_bg_open`: an update has put work on the side stream that the current stream has not waited for), so `engine.params[...]`,
    `engine.flat_m.zero_()` or a `.cpu()` after optim_step() see -- and write behind -- the finished update in every mode"""
    priv = "_" + name

    def get(self):
        dd = self.__dict__
        if not dd.get("_in_flush"):
            if dd.get("_pending_bg") or (dd.get("_bg_open") and not dd.get("_in_step")):
                self.wait_background()
            if dd.get("_lazy_dirty") and not dd.get("_in_step"):
                self.flush_lazy_rows()          # (the embedding tables' rows are brought up to date lazily: _build_row_tables)
        try:
            return dd[priv]
        except KeyError:
            raise AttributeError(name)

    def put(self, value):
        self.__dict__[priv] = value
    return property(get, put)


class Engine(StepAPI):
    params = _behind_background("params")
    flat_p = _behind_background("flat_p")
    flat_m = _behind_background("flat_m")
    flat_v = _behind_background("flat_v")

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
        self._step_count = 0         # Adam step counter (property step_count: the lazy embedding tables follow a value set from outside)
        self._adam_consts = (0.9, 0.999, 1e-9)      # beta1, beta2, eps of the run (Optim.py:68-70); optim_step refuses to change them under lazy rows
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
        self.use_aux_stream = True
        # fourth stream (conditional model only): encoder_tgt's backward recurrence, 2 x B dependent steps that would otherwise hold
        # up everything queued behind them on the aux stream
        # (created on first use -- two-layer / conditional models only: every stream is one more hardware queue for the GPU's scheduler to
        #  rotate, and with six of them the persistent recurrences' launches were seen waiting 0.2-0.4 ms for their queue to be mapped)
        self._lo_pri, self._hi_pri = lo_pri, hi_pri
        self._tgt_stream = None
        # workgroups of the BACKGROUND half of Adam (side stream, underneath the next step's encoder recurrence).  Unthrottled it takes
        # the memory system for ~200 us and the latency-bound persistent LSTM kernel next to it runs at half speed; one workgroup per
        # CU still finishes before the decoder-side weights are needed.  tools/ab.py, ms per step: 4096 wgs 1.958-2.007 | 384: 1.957 |
        # 288: 1.972 | 256: 1.921-1.951 | 224: 1.908 | 192: 1.987 | 128: 2.051 | 64: 2.324 (a faster two-chunk kernel at 256: 1.98)
        self.bg_adam_blocks = 256
        # ... and of the FOREGROUND half (0 = uncapped, 4096).  Alone on the chip the kernel streams 5.6 TB/s uncapped and 6.0 TB/s from 512
        # workgroups (tools/hbm_kernels.py: 108.5 -> 101.1 us for the 21.8 M parameters of that half)
        self.fg_adam_blocks = 0
        # cap of the weight-gradient products' split-K.  On an idle chip 4, 8 and 16 splits cost the same (tools/gemm_split.py), in the step
        # the extra workgroups and atomics get in the way of everything that runs next to them (tools/ab.py, ms per step by cap: 64:
        # 1.854, 8: 1.834-1.843, 6: 1.798, 5: 1.782, 4: 1.792-1.812, 3: 1.815, 2: 1.848, 1: 2.003)
        self.max_split_k = 4
        self.wgrad_target_tiles = 1024     # 64 x 64 output tiles x splits a weight-gradient product aims at
        self.cond_aux_early = True
        self.cond_emb_fg = True
        self.aux_early = True
        self.aux_kl_first = True
        self.gen_db_in_gemm = True
        self.split_combine = True    # the sweep's fold in two launches: the dWg product starts beside the fold of dO, not behind it
        self.lstm_db_in_gemm = True
        self.dec_grads_on_aux = True
        self.bwd_main_first = True      # issue order of the backward plan (see _plan_backward)
        self.bwd_layers_parallel = True   # >= 2 layers: top encoder layer next to the lower decoder layers
        # the weight-gradient products of a layer (dW_hh / dW_ih of each direction; attention's two) as ONE grid each (vmmt_gemm_group)
        # -- not with the --conditional prior: there the step's length is encoder_tgt's chain (2 x B = 512 sequential LSTM steps at batch 20,
        # 0.86 + 1.5 ms) and every weight gradient runs underneath it; a grouped grid takes more of the chip at once and slows that chain
        # (same box, three pairs: 3.73 / 3.72 / 3.80 ms grouped against 3.47 / 3.45 / 3.49 ms solo; profiles/r4_conditional_grouping.txt)
        self.group_wgrads = _os_env.get("VMMT_GROUP_WGRADS", "0" if dims.conditional else "1") == "1"
        # (a high-priority stream for the critical path was measured and is slightly SLOWER than the default stream:
        #  tools/sched_ab.py, 3.249 vs 3.226 ms/step)
        self._compute_stream = None
        self._plan_tgt = {}           # id(plan) -> (entries, whether it uses the fourth stream)
        self.use_side_stream = True
        # the generator's calls over the decoder rows that carry a target only (pads compacted away) when forward() knows their number
        self.gen_compact = True
        self._masked_streams = []
        import os as _os
        self.q_parallel = True    # q(z|x): scale branch on the side stream next to the location branch
        # 2: the FOREGROUND half of the update (the step's tail) as one dense launch around its lazily updated table + the table's row-wise one
        # (config 2: 1.438-1.443 against 1.449-1.456 ms per piece; the background half the same way: 1.472-1.476 -- its pieces stay; 1: both, 0: neither)
        self.adam_one_launch = 2
        self.tail_norm_first = True   # backward plan: the encoder segment's norm in front of the main stream's join with the aux stream
        self.dec_gx_first = True  # side stream of the forward: the decoder's input projection in front of the gradient zeroing / masks
        self.zero_on_aux = True   # ... and the zeroing / masks on the AUX stream (idle at that point of the forward) instead of behind it
        self.fuse_emb_gemm = True # the source embeddings as the gathered A operand of the encoder's first input projection (Engine.row_shadow)
        self.trace = None            # list -> _run appends (name, timing event) at every main-stream phase change
        self._kl_sum_needed = True   # set per step by Workspace.backward_plan: does the latent backward read the (global) KL sum?
        self.trace_only = None       # ... of these entry names only (every event costs the stream ~10 us: tools/phase_times.py COARSE=1)
        self.global_events = {}      # events that outlive a plan run (optimizer <-> next forward)
        self.split_optim = True      # run the decoder-side half of Adam + shadow refresh on the side stream
        # the side-stream half of Adam held back until the NEXT forward's head (source gather + the encoder's first input projection) is
        # through: next to 200 us of optimiser streaming that product takes 112 instead of 30 us (profiles/r4_step_timeline.txt)
        # Default: models of two or more layers (scripts' shape at batch 40 / 256: 1.86 -> 1.73 / 2.68 -> 2.55 ms, config 5: 17.49 -> 17.26;
        # config 2 with its one encoder layer: 1.738 against 1.741 -- its encoder phase is shorter than that half of the update).  The plans
        # are laid out for it when this is set; an update is only HELD BACK while `hold_back` is set, which the owner of a training loop
        # does (TrainerMultimodal._train_loop, bench.py) and clears at the loop's end: outside such a loop optim_step() issues everything
        bah = _os_env.get("VMMT_BG_AFTER_HEAD", "auto")
        self.bg_after_head = (dims.layers >= 2) if bah == "auto" else bah == "1"
        self.hold_back = _os_env.get("VMMT_HOLD_BACK", "0") == "1"      # (the environment switch is for bug hunts: the whole test suite held back)
        self._pending_bg = None
        self._sumsq = torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=self.dev)   # slot totals | tickets | partials (vmmt.h)
        self._sumsq_by_plan = False
        self._normed = set()         # sharded data parallelism: arena segments whose shard the backward plan has normed (step.py)
        self.fused_qnet = True
        self.qnet_split = True     # location / scale networks in separate workgroups (csrc/qnet.hip)
        self.gen_fused = _os_env.get("VMMT_GEN_FUSED", "1") == "1"       # csrc/generator_fused.hip where it applies (bf16, H = 512 / 256)
        self.persistent_lstm = _os_env.get("VMMT_PERSISTENT_LSTM", "1") == "1"     # plans are built per workspace: set before the first forward
        self.seq_syncs = []
        # GUARD word of the persistent recurrences (vmmt.h: VMMT_SEQ_GUARD_WORD): [0] error code of a hand-off that timed out (sticky),
        # [1] optimiser launches skipped because of it.  vmmt_adam_step reads it on the device, the host sees a pinned copy one or two
        # steps later (optim_step) and then continues on the per-step kernels (_seq_timeout_fallback)
        self._guard = torch.zeros(2, dtype=torch.int32, device=self.dev)
        # the host's view of it: a ring of three pinned copies, one written behind every optimiser step, each with its event.  One process
        # reads whatever has arrived; data-parallel ranks all read the copy of TWO steps ago (waiting for its event: it has long
        # arrived) -- the folded guard is the same number on every rank, so every rank switches over, skips and clears on the same step
        self._guard_host = torch.zeros(3, 2, dtype=torch.int32, device="cpu").pin_memory()      # (explicit device: a driver may have made CUDA the default tensor type)
        self._guard_events = [torch.cuda.Event() for _ in range(3)]
        self._guard_pub = 0
        self.seq_fallback = _os_env.get("VMMT_SEQ_FALLBACK", "1") == "1"          # 0: a timeout raises (check_async_errors) as before round 4
        self.seq_fallbacks, self.steps_skipped, self._adam_launches, self._guard_clear_pending = 0, 0, 1, False
        self.dp = None               # dp.GradSync when torch.distributed runs with > 1 rank
        self._works = []
        # ONE switch for experiments instead of one per schedule knob: VMMT_ENGINE_ATTRS="bg_adam_blocks=224;aux_early=0" sets schedule
        # attributes of this object (what tools/ab.py sets programmatically) before the first plan is built.  Only the knobs listed
        # in _ATTR_KNOBS, each with its parser; pairs are separated by ';' (or ',': tools/ab_env.sh separates its ARMS by commas)
        for kv in filter(None, _os_env.get("VMMT_ENGINE_ATTRS", "").replace(";", ",").split(",")):
            k, _, v = (x.strip() for x in kv.partition("="))
            if k not in self._ATTR_KNOBS:
                raise RuntimeError("VMMT_ENGINE_ATTRS: %r is not a schedule knob of the engine (knobs: %s)" % (k, ", ".join(sorted(self._ATTR_KNOBS))))
            try:
                setattr(self, k, self._ATTR_KNOBS[k](v))
            except ValueError:
                raise RuntimeError("VMMT_ENGINE_ATTRS: %s=%r does not parse as %s" % (k, v, self._ATTR_KNOBS[k].__name__))
        gone = sorted(n for n in _os_env if n in self._REMOVED_SWITCHES)
        if gone:
            import sys
            print("[vmmt] WARNING: %s no longer exist as environment switches and are IGNORED; the schedule knobs are attributes now: "
                  "VMMT_ENGINE_ATTRS=\"name=value;...\" (DESIGN.md section 10)" % ", ".join(gone), file=sys.stderr, flush=True)

    def _knob_bool(v):
        if v.lower() in ("1", "true", "on", "yes"):
            return True
        if v.lower() in ("0", "false", "off", "no", ""):
            return False
        raise ValueError(v)
    _knob_bool.__name__ = "bool"
    # the schedule knobs an experiment may set through VMMT_ENGINE_ATTRS (name -> parser); everything else is refused
    _ATTR_KNOBS = dict(
        bg_adam_blocks=int, fg_adam_blocks=int, max_split_k=int, wgrad_target_tiles=int, lazy_roll=int, shape_bucket=int,
        cond_aux_early=_knob_bool, cond_emb_fg=_knob_bool, aux_early=_knob_bool, aux_kl_first=_knob_bool, gen_db_in_gemm=_knob_bool, split_combine=_knob_bool,
        lstm_db_in_gemm=_knob_bool, dec_grads_on_aux=_knob_bool, bwd_main_first=_knob_bool, bwd_layers_parallel=_knob_bool,
        group_wgrads=_knob_bool, use_side_stream=_knob_bool, use_aux_stream=_knob_bool, gen_compact=_knob_bool, q_parallel=_knob_bool,
        dec_gx_first=_knob_bool, zero_on_aux=_knob_bool, fuse_emb_gemm=_knob_bool, split_optim=_knob_bool, bg_after_head=_knob_bool, hold_back=_knob_bool,
        fused_qnet=_knob_bool, qnet_split=_knob_bool, gen_fused=_knob_bool, persistent_lstm=_knob_bool, seq_fallback=_knob_bool,
        row_adam=_knob_bool, reparam_grad=_knob_bool, tail_norm_first=_knob_bool, adam_one_launch=int)
    # environment switches of rounds 2-4 that became attributes (or went with their kernels) in round 5: setting one is a mistake worth a line
    _REMOVED_SWITCHES = frozenset((
        "VMMT_LATENT_ZX", "VMMT_FUSE_OUT_DROPOUT", "VMMT_DECODE_GRAPHS", "VMMT_BG_ADAM_BLOCKS", "VMMT_FG_ADAM_BLOCKS", "VMMT_MAX_SPLIT_K",
        "VMMT_WGRAD_TARGET_TILES", "VMMT_AUX_EARLY", "VMMT_AUX_KL_FIRST", "VMMT_COND_AUX_EARLY", "VMMT_COND_EMB_FG", "VMMT_COND_EMB_FIRST",
        "VMMT_COND_DEC_STEPS", "VMMT_GEN_DB_IN_GEMM", "VMMT_LSTM_DB_IN_GEMM", "VMMT_DEC_GRADS_ON_AUX", "VMMT_BWD_MAIN_FIRST",
        "VMMT_BWD_LAYERS_PARALLEL", "VMMT_QPAR", "VMMT_QNET_SPLIT", "VMMT_FUSED_QNET", "VMMT_AUX_STREAM", "VMMT_SPLIT_OPTIM", "VMMT_GEN_COMPACT",
        "VMMT_FUSE_DB", "VMMT_DEFER_BG_ADAM", "VMMT_PAD_HIDDEN", "VMMT_LAZY_EMB_ADAM"))

    @property
    def use_side_stream(self):
        return self._use_side_stream

    @use_side_stream.setter
    def use_side_stream(self, on):
        """changed between steps (bench.py's overlap self-check, a profiling run): a held-back half of the last update goes out first and the
        device drains -- in single-stream mode the plans skip every cross-stream wait, `opt_side_done` included, so work still running on
        the side / aux streams would race the next step's main-stream kernels"""
        on = bool(on)
        if "_use_side_stream" in self.__dict__ and on != self._use_side_stream and hasattr(self, "ws"):
            self.wait_background()
            torch.cuda.synchronize(self.dev)
        self._use_side_stream = on

    @property
    def tgt_stream(self):
        if self._tgt_stream is None:
            self._tgt_stream = torch.cuda.Stream(device=self.dev, priority=self._lo_pri)
        return self._tgt_stream

    @tgt_stream.setter
    def tgt_stream(self, s):
        self._tgt_stream = s

    @property
    def compute_stream(self):        # (tools/ab.py only)
        if self._compute_stream is None:
            self._compute_stream = torch.cuda.Stream(device=self.dev, priority=self._hi_pri)
        return self._compute_stream

    @compute_stream.setter
    def compute_stream(self, s):
        self._compute_stream = s

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
        """EXACT LAZY Adam for the two embedding tables (csrc/optim.hip, include/vmmt.h: vmmt_rows_mark / _catchup / vmmt_sumsq_rows /
        vmmt_adam_rows_step).  The tables are 54 % of the optimised parameters and a step's gradient lives in the <= S B + T' B rows
        the batch looked up (17 % of 30 000 at the benchmark shape), yet torch.optim.Adam (Optim.py:68-70,94-96) -- and the dense
        kernels here -- clear, norm and stream all of g, p, m, v at every step: 36 B per element, 1.08 of the step's 2.2 GB.  A row
        without gradient still moves under Adam (its moments decay, the parameter follows them), but that zero-gradient step needs
        nothing but the row's own state and the step's scalars: it is applied LATER with the dense kernel's arithmetic in the dense
        kernel's order, so parameters and moments stay bit-identical to the dense update (tests/test_gpu_row_adam.py):
          * the training forward flags the batch's rows and brings them up to date in front of the lookup (mark + catch-up), clearing
            their gradient rows for the backward's scatter-add;
          * the norm and the update touch the flagged rows -- and a ROLLING 1 / lazy_roll of each table per step, so that no row is
            ever more than lazy_roll steps behind: the replay in front of a lookup is a few microseconds whatever the distribution of
            the word ids.  (Rounds 1-3 had the lazy update without that bound: with Zipf-distributed ids the rare words' rows came
            back after hundreds of steps and their replays, in front of the lookup, cost more than the traffic saved.)
          * everything that reads the tables from outside a training step -- state_dict(), `engine.params / flat_p / flat_m / flat_v`,
            evaluation and decoding plans, checkpoints, the dense optimisers, data parallelism -- flushes first (every row up to date).
        Off under data parallelism (the flagged set would have to be the union over the ranks).  The conditional model's encoder_tgt looks
        the SHARED target table up as well (ModelConstructor.py:456-457): its ids are flagged by a launch of their own in front of the
        table's one catch-up, which both lookups wait for (Workspace._plan_forward); off by default there (no gain measured).
        `Engine.row_adam = False` restores the dense kernels."""
        names = ("encoder.embeddings.make_embedding.emb_luts.0.weight", "decoder.embeddings.make_embedding.emb_luts.0.weight")
        self._row_adam = False
        self._lazy_dirty = False             # rows behind the step counter exist (an update has run since the last flush)
        self._in_step = 0                    # > 0: inside forward / loss_backward / optim_step / a flush (reads of the arena are the engine's own)
        self.lazy_roll = 16                  # every row is brought up to date at least every `lazy_roll` updates (<= LAZY_HIST / 4)
        self.row_tables = []
        for n in names:          # (the bookkeeping arrays are always there -- 0.5 MB -- so that the switch can be set after construction)
            off, (R, Cc) = self.offsets[n]
            if Cc % 4 or off % 4:
                self._row_adam, self.row_tables = False, []
                return
            self.row_tables.append(dict(name=n, off=off, R=R, C=Cc, end=off + R * Cc,
                                        flags=torch.zeros(2 * R, dtype=torch.int32, device=self.dev),      # [parity of the update][row]
                                        last=torch.zeros(R, dtype=torch.int32, device=self.dev),
                                        hist=torch.zeros(L.LAZY_HIST_WORDS, dtype=torch.int32, device=self.dev),
                                        rowsq=torch.zeros(R, dtype=torch.float32, device=self.dev), shadow=None))
        # (the conditional model: built and tested -- tests/test_gpu_row_adam.py -- but level with the dense update there, 2.655-2.666 against
        #  2.635-2.657 ms: the catch-up stands at the head of encoder_tgt's chain, that step's longest, for what the update's tail saves.
        #  `engine.row_adam = True` switches it on)
        self._row_adam = _os_env.get("VMMT_ROW_ADAM", "1") == "1" and not self.d.conditional

    @property
    def row_adam(self):
        return self._row_adam

    @row_adam.setter
    def row_adam(self, on):
        """the switch may be set after construction: the cached launch plans carry (or omit) the row entries; switched OFF, every row is
        brought up to date first and the dense kernels find fully cleared table gradients; switched ON, every row is current for the
        step counter as it stands"""
        on = bool(on) and bool(self.row_tables)
        if on == getattr(self, "_row_adam", False):
            return
        if hasattr(self, "ws"):
            self.flush_lazy_rows()          # (no-op unless the lazy path has run)
            self._flush_bg()
            self.drop_workspaces()
        self._row_adam = on
        if hasattr(self, "ws"):
            self._lazy_reset(self._step_count)
            for t in self.row_tables:
                self.flat_g[t["off"]:t["end"]].zero_()

    def _lazy_reset(self, n):
        """every row of the tables is current for step n (fresh moments, a loaded optimiser state, after a flush)"""
        for t in self.row_tables:
            t["last"].fill_(int(n))
            t["flags"].zero_()
            t["hist"].zero_()
            t["hist"][0] = int(n)
        self._lazy_dirty = False

    @property
    def step_count(self):
        """Adam's step counter (Optim.py: the optimiser's `state['step']`)"""
        return self._step_count

    @step_count.setter
    def step_count(self, n):
        """set from outside: a new Adam (0), a loaded optimiser state, a twin engine in a test.  The lazy tables follow: rows are
        brought up to date for the OLD counter first, then declared current for the new one.  (The guard's roll-back of skipped steps
        goes through _step_rollback: those steps moved nothing, a row's `last` is clamped instead.)"""
        n = int(n)
        if getattr(self, "row_tables", None) and hasattr(self, "ws"):
            self.flush_lazy_rows()
            self._lazy_reset(n)
        self._step_count = n

    def _step_rollback(self, n):
        """the step counter goes BACK to n: the updates n + 1 .. were skipped on the device (guard word) and are numbered again.  They
        are recorded as skipped in the tables' rings and moved nothing: a row that is 'current' for one of them is current for n"""
        n = max(0, int(n))
        if self.row_tables and n < self._step_count:
            for t in self.row_tables:
                t["last"].clamp_(max=n)
                t["hist"][0] = n
        self._step_count = n

    def dp_on(self):
        """collectives are part of the step: torch.distributed with > 1 rank (or the forced one-rank rehearsal, dp.GradSync)"""
        return self.dp is not None and self.dp.active()

    def rows_active(self):
        return bool(self.row_tables) and self.row_adam and not self.dp_on()

    def flush_lazy_rows(self, stream=None):
        """every row of the embedding tables up to date for the step counter (vmmt_rows_catchup, mode 1), on the current stream behind
        both halves of the last update.  Called by everything that reads the tables outside a training step; no-op when nothing is behind."""
        if not self.__dict__.get("_lazy_dirty") or not self.row_tables:
            return
        self._in_step += 1
        try:
            self._lazy_dirty = False
            self.wait_background()
            st = torch.cuda.current_stream(self.dev).cuda_stream
            b1, b2, eps = self._adam_consts
            for t in self.row_tables:
                o = 4 * t["off"]
                L.check(self.lib.vmmt_rows_catchup(self.flat_p.data_ptr() + o, None, self.flat_m.data_ptr() + o, self.flat_v.data_ptr() + o,
                                                   t["R"], t["C"], None, t["last"].data_ptr(), t["hist"].data_ptr(), b1, b2, eps, 1, st),
                        "vmmt_rows_catchup")
        finally:
            self._in_step -= 1

    def lazy_errors(self):
        """error words of the tables' rings (0 = no replay ever met an overwritten entry); synchronises"""
        return [int(t["hist"][1].item()) for t in self.row_tables]

    def _row_flag_args(self, training):
        """the row-flag arguments of vmmt_prepare_batch: a training batch flags the rows it looks up in both tables for the NEXT update
        (the step counter is the host's: equal to the tables' own `hist[0]` except inside _step_rollback, which runs with the device idle)"""
        if not (training and self.rows_active()):
            return (None, 0, None, 0, 0)
        s_, t_ = self.row_tables
        return (s_["flags"].data_ptr(), s_["R"], t_["flags"].data_ptr(), t_["R"], self._step_count + 1)

    def _row_mark_entries(self, plan, table_index, ids_ptr, n_ids, with_shadow=False):
        """plan entry (training forward, IN FRONT of the table's lookup): the rows the batch flagged (vmmt_prepare_batch) are brought up to
        date and their gradient rows cleared, which the backward plan's scatter-add accumulates into"""
        if not self.rows_active():
            return
        t = self.row_tables[table_index]
        o = 4 * t["off"]
        b1, b2, eps = self._adam_consts
        sh = self.row_shadow(table_index) if with_shadow else None
        if sh is not None:
            self._call(plan, self.lib.vmmt_rows_catchup_shadow, self.flat_p.data_ptr() + o, self.flat_g.data_ptr() + o, self.flat_m.data_ptr() + o,
                       self.flat_v.data_ptr() + o, t["R"], t["C"], t["flags"].data_ptr(), t["last"].data_ptr(), t["hist"].data_ptr(),
                       b1, b2, eps, sh.p(), sh.ld)
            return
        self._call(plan, self.lib.vmmt_rows_catchup, self.flat_p.data_ptr() + o, self.flat_g.data_ptr() + o, self.flat_m.data_ptr() + o,
                   self.flat_v.data_ptr() + o, t["R"], t["C"], t["flags"].data_ptr(), t["last"].data_ptr(), t["hist"].data_ptr(),
                   b1, b2, eps, 0)

    def row_shadow(self, table_index):
        """bf16 copy of an embedding table with 16-byte aligned rows (padded to whole K slabs, the padding zero), CURRENT FOR THE ROWS OF THE
        BATCH ONLY: the catch-up of a training forward writes a row's copy behind bringing it up to date.  The table the encoder's first
        input projection fetches its A operand from by token id (vmmt_gemm_args.a_row_ids): north_star's embedding lookup fused into the
        LSTM gate GEMM.  None where that does not apply (fp32, the dense update)"""
        if not (self.rows_active() and self.dt == L.BF16 and self.fuse_emb_gemm):
            return None
        t = self.row_tables[table_index]
        if t["shadow"] is None:
            t["shadow"] = Buf(t["R"] + 64, t["C"], torch.bfloat16, self.dev)       # (+ rows of slack behind the last one)
        return t["shadow"]

    def pp(self, name, r=0, c=0):
        o, shp = self.offsets[name]
        ld = shp[1] if len(shp) > 1 else 0
        return self.flat_p.data_ptr() + (o + r * ld + c) * 4

    def gp(self, name, r=0, c=0):
        o, shp = self.offsets[name]
        ld = shp[1] if len(shp) > 1 else 0
        return self.flat_g.data_ptr() + (o + r * ld + c) * 4

    def wait_background(self, stream=None):
        """`stream` (default: the current one) waits for the half of the last optimiser step that runs on the side stream; called from
        outside a step (whoever is about to read parameters), the lazily updated embedding rows are brought up to date as well"""
        if self.__dict__.get("_lazy_dirty") and not self._in_step:
            self.flush_lazy_rows()
        self._flush_bg()
        for name in ("opt_side_done", "opt_gen_done"):
            ev = self.global_events.get(name)
            if ev is not None:
                (stream if stream is not None else torch.cuda.current_stream(self.dev)).wait_event(ev)
        if stream is None:
            self._bg_open = False

    def _flush_bg(self, after=None, parts=None, stream=None):
        """issue the side-stream half of the last optimiser step if it was held back (bg_after_head): its first `parts` pieces (default:
        all that are left); `after`: a stream whose work issued so far it must stay behind"""
        bg = self._pending_bg
        if bg:
            if after is not None:
                ev = self.global_events.setdefault("bg_head", torch.cuda.Event())
                ev.record(after)
                self.side_stream.wait_event(ev)
            self._in_flush = True
            try:
                for _ in range(len(bg) if parts is None else min(parts, len(bg))):
                    bg[0](stream)              # (popped when it has gone out: a part that raises is neither lost nor left half-issued
                    bg.pop(0)                  #  behind a later part -- the error surfaces, the update is still whole on the next flush)
            finally:
                self._in_flush = False
        if not bg:
            self._pending_bg = None

    def _bg_cut_ok(self):
        """may the generator's third of the arena be updated BEHIND the shadow refresh of the side-stream half?  Only if that refresh reads
        nothing of it (the generator weight's shadow is written by the update itself in the bf16 layouts: _fused_shadows)"""
        if not hasattr(self, "_bg_cut"):
            base, lim = self.flat_p.data_ptr(), self.flat_p.data_ptr() + 4 * self.segments[1][0]
            fused = set(d for _, _, d in self._fused_shadows())
            enc_lo = self.offsets[self.first_enc_name][0]
            sel = [c for c in self.pack_calls if (c[1] - base) // 4 < enc_lo and c[4] not in fused]
            self._bg_cut = all(not (base <= q < lim) for c in sel for q in (c[1], c[2]) if q)
        return self._bg_cut

    def load_state_dict(self, sd):
        self.wait_background()
        for n, t in sd.items():
            if n in self.params:
                self.params[n].copy_(t.to(torch.float32))
        self.shadows_dirty = True

    def state_dict(self):
        self.wait_background()      # (decoder-side parameters are updated on the side stream: optim_step)
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
              b_batch_rows=0, b_batch_stride=0, colsum=None, rmap=None, cmap=None, group=None, a_row_ids=None):
        """plan entry: one vmmt_gemm.  colsum = (w, w_stride, out[, out2]): the column sums of the K-strided A operand from the same
        pass (weighted by w, or plain with w = None), where the library offers them; returns whether they were attached.
        group (a list): the product is not issued but collected -- _gemm_group() sends the list out as ONE grid (vmmt_gemm_group)"""
        if split_k == -1:
            # weight-gradient heuristic: enough workgroups to fill 256 CUs, >= 256 reduction steps each, at most max_split_k splits
            tiles = ((M + 63) // 64) * ((N + 63) // 64)
            split_k = max(1, min(K // 256, (self.wgrad_target_tiles + tiles - 1) // tiles, int(self.max_split_k)))
            if group is not None and K >= 128:
                if split_k == 1 and 2.0 * M * N * K > 8e9:
                    # a product that fills the chip by itself (config 5: [4096 x 1024 x 16384]) keeps its plain store -- as a member of a
                    # group it would have to accumulate with atomics (17.65 -> 17.85 ms per step): the group is issued one by one then
                    group.append(None)
                else:
                    split_k = max(2, split_k)   # members of a grouped launch accumulate with atomics (several may share one C)
            if split_k == 1:
                accumulate = 1          # gradients always ACCUMULATE into the arena (zeroed at the start of a step)
        if a_kmod == 0 and b_kmod == 0:
            K = _ru(K, KPAD)            # operands are Bufs: zero-padded to whole slabs (see Buf)
        a = L.GemmArgs(self.dt, layout, A, lda, B, ldb, Cp, ldc, M, N, K, a_kmod, b_kmod, addend, ld_add, add_rows,
                       add_is_T, act, out_f32, accumulate, alpha, scatter_ids, PAD, tile, split_k, b_batch_rows, b_batch_stride,
                       None, 0, None, None)
        a.a_row_ids = a_row_ids       # (NT: row m of A is row a_row_ids[m] of the table at A -- the embedding lookup as the product's operand fetch)
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
        if group is not None:
            group.append(a)
            return attached
        plan.append((self.lib.vmmt_gemm, (C.byref(a),), "gemm", a, self._sid))
        return attached

    def _gemm_group(self, plan, group):
        """plan entry: the products collected in `group` (_gemm(group=)) as one vmmt_gemm_group call -- one grid where the library
        groups them (bf16 weight-gradient products on 128 x 128 tiles), one launch per product otherwise"""
        if not group:
            return
        solo = any(a is None for a in group)         # (a member asked to stay on its own: see _gemm)
        group[:] = [a for a in group if a is not None]
        if solo or len(group) == 1 or len(group) > 8:
            for a in group:
                plan.append((self.lib.vmmt_gemm, (C.byref(a),), "gemm", a, self._sid))
            return
        arr = (L.GemmArgs * len(group))()
        for i, a in enumerate(group):
            C.memmove(C.byref(arr, i * C.sizeof(L.GemmArgs)), C.byref(a), C.sizeof(L.GemmArgs))
        plan.append((self.lib.vmmt_gemm_group, (arr, len(group)), "gemm_group", arr, self._sid))

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
        """A persistent recurrence needs all of its workgroups resident at once: (B / 16 row groups) x (H / 32 unit slices) x directions
        <= 256.  Sentences are independent in a recurrence, so a batch that does not fit is cut into ROW chunks, one persistent launch
        each, one after the other (BASELINE config 5: H = 1024 -> 32 slices -> 128 sentences per launch; the per-step kernels it
        replaces re-read their W_hh slice from L2 at every step: 35 us per backward step against 8).  -> [(descriptors, row offset,
        rows)], or None when the persistent kernel does not serve this size at all."""
        if H not in (64, 128, 256, 512, 1024):
            return None
        groups = 256 // ((H // 32) * ndir)
        if groups < 1:
            return None
        rows = 16 * groups
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
        persistent kernel (W_hh resident in registers, in-launch hand-off of h_t: csrc/lstm_seq.hip), which falls back by itself to the
        per-step kernels where it does not apply; otherwise the per-step kernels issued from one host call."""
        chunks = self._seq_row_chunks(arr, self._SEQ_F, ndir, B, H) if (self.persistent_lstm and self.dt == L.BF16) else None
        if chunks is not None and len(chunks) > 1:
            for sub, r0, rows in chunks:
                self._lstm_seq_fwd(plan, sub, ndir, nsteps, (lens_ptr + 8 * r0) if lens_ptr else None, rows, H)
            return
        if self.persistent_lstm:
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
            sync = self._seq_sync()
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
            sync = self._seq_sync()
            xchg = torch.zeros(max(16, self.lib.vmmt_lstm_seq_xchg_bytes_bwd(ndir, B, H)), dtype=torch.uint8, device=self.dev)
            plan.append((self.lib.vmmt_lstm_seq_bwd, (self.dt, ndir, nsteps, arr, dev.data_ptr(), lens_ptr, B, H, with_dh0, sync.data_ptr(),
                                                      xchg.data_ptr()), "vmmt_lstm_seq_bwd", (arr, dev, sync, xchg), self._sid))
            self.seq_syncs.append(sync)
        else:
            plan.append((self.lib.vmmt_lstm_chain_bwd, (self.dt, ndir, nsteps, arr, lens_ptr, B, H, 0), "vmmt_lstm_chain_bwd", arr, self._sid))
            if with_dh0:
                last = C.cast(C.byref(arr, nsteps * ndir * C.sizeof(L.LstmDirBwd)), C.POINTER(L.LstmDirBwd))
                plan.append((self.lib.vmmt_lstm_step_bwd, (self.dt, ndir, last, lens_ptr, B, H, 1), "vmmt_lstm_step_bwd", arr, self._sid))

    def _seq_sync(self):
        """sync words of one persistent launch site, the engine's guard pointer in their last words"""
        sync = torch.zeros(self.lib.vmmt_lstm_seq_sync_words(), dtype=torch.int32, device=self.dev)
        sync.view(torch.int64)[L.SEQ_GUARD_WORD // 2] = self._guard.data_ptr()
        return sync

    def lstm_seq_errors(self):
        """error words of the persistent recurrence launches so far (0 = every in-launch wait completed); synchronises"""
        torch.cuda.synchronize(self.dev)
        return [int(s[2].item()) for s in self.seq_syncs]            # [launch epoch, finish count, error word]

    def _seq_timeout_fallback(self, where, in_step=False):
        """An in-launch wait of a persistent recurrence ran into its 2-second bound (a workgroup of the row group was not resident:
        another process on the GPU, a CU mask, collective kernels holding CUs).  The device has already protected the model -- every
        optimiser launch since then saw the guard word and changed nothing -- so the run CONTINUES on the per-step kernels (one launch
        per time step: slower, no residency requirement) and says so, instead of aborting the job."""
        import sys
        self._flush_bg()
        torch.cuda.synchronize(self.dev)
        codes = [w for w in (int(s[2].item()) for s in self.seq_syncs) if w]
        g = self._guard.tolist()
        skipped = g[1] // max(1, self._adam_launches)
        # (-optim sgd|adagrad|adadelta: torch's dense optimisers do not read the guard -- the steps since the timeout may have applied
        #  garbage gradients: nothing to continue from)
        if not self.seq_fallback or getattr(self, "dense_optimizer", False):
            raise RuntimeError("persistent LSTM launch(es) reported a hand-off timeout (error words %s, guard 0x%x): results invalid; "
                               "rerun with VMMT_PERSISTENT_LSTM=0" % ([hex(c) for c in codes], g[0]))
        self.persistent_lstm = False
        self.seq_syncs = []
        self.drop_workspaces()
        self._step_rollback(self._step_count - skipped)           # Adam's step counter: those updates never happened
        if in_step:
            # called from optim_step: the step at hand ran its recurrences on the persistent kernels as well (they may have timed out
            # too, and the synchronisation above has just let them finish): the guard stays set through THIS update -- skipped on the
            # device like the ones before it, consistently on every data-parallel rank -- and is cleared behind it (optim_step)
            self._guard_clear_pending = True
            skipped += 1
        else:
            self._guard.zero_()
        self._guard_host.zero_()
        self.seq_fallbacks += 1
        self.steps_skipped += skipped
        print("[vmmt] WARNING (%s): a persistent LSTM recurrence timed out waiting for its row group (error words %s, guard 0x%x); "
              "%d optimiser step(s) were skipped on the device, nothing wrong was applied.  Continuing with one launch per time step "
              "(VMMT_PERSISTENT_LSTM=0) for the rest of the run." % (where, [hex(c) for c in codes], g[0], skipped), file=sys.stderr, flush=True)

    def poll_guard(self, where="optim_step"):
        """host side of the guard: a pinned copy of the word, refreshed asynchronously behind every optimiser step -- reading it costs no
        synchronisation; the first step that sees it set switches the engine over"""
        if self.dp_on():
            # every rank publishes once and polls once per optimiser step: the same slot, the same (folded) value, the same decision
            if self._guard_pub < 2:
                return False
            k = (self._guard_pub - 2) % 3
            self._guard_events[k].synchronize()
            seen = int(self._guard_host[k, 0]) != 0
        else:
            seen = bool((self._guard_host[:, 0] != 0).any())
        if seen:
            self._seq_timeout_fallback(where, in_step=True)
            return True
        return False

    def _publish_guard(self, stream):
        """end of an optimiser step, on the stream that carries its last Adam launch: clear the guard if the host has just dealt with it,
        refresh the host's pinned copy"""
        if getattr(self, "_guard_clear_pending", False):
            with torch.cuda.stream(stream):
                self._guard.zero_()
            self._guard_clear_pending = False
            # (the update that was skipped with the guard still set.  Rare: the lazy tables' bookkeeping goes back with the device idle)
            torch.cuda.synchronize(self.dev)
            self._step_rollback(self._step_count - 1)
            torch.cuda.synchronize(self.dev)
        with torch.cuda.stream(stream):
            k = self._guard_pub % 3
            self._guard_host[k].copy_(self._guard, non_blocking=True)
            self._guard_events[k].record(stream)
            self._guard_pub += 1

    def check_async_errors(self):
        """Synchronises and settles everything the device reports asynchronously: a persistent recurrence that timed out (-> the engine
        falls back to the per-step kernels with a warning, _seq_timeout_fallback; VMMT_SEQ_FALLBACK=0: raises) and a step that was told
        fewer target tokens than its batch held.  The trainer mirror calls it at the end of every epoch and before a checkpoint is
        written, bench.py after its timed region."""
        self._flush_bg()            # (a held-back half of the last update reads the guard: it goes out before the guard is settled)
        torch.cuda.synchronize(self.dev)
        if int(self._guard[0].item()) != 0 or any(self.lstm_seq_errors()):
            self._seq_timeout_fallback("check_async_errors")
        # ... or if a step was told fewer target tokens than its batch held (forward(n_tgt_tokens=)): the generator left rows out
        # ... or if a replay of the lazily updated embedding tables met an overwritten entry of its ring of step scalars (a row further behind
        # than VMMT_LAZY_HIST updates: the rolling update bounds that by lazy_roll; only a caller driving the kernels past it gets here)
        if any(self.lazy_errors()):
            raise RuntimeError("lazy embedding tables: a row was more than %d optimiser steps behind (ring overrun at step(s) %s): its "
                               "parameters are wrong; lower Engine.lazy_roll or set Engine.row_adam = False" % (L.LAZY_HIST, self.lazy_errors()))
        short = [k for k, w in self.ws.items() if isinstance(w, Workspace) and getattr(w, "gen_cnt", None) is not None and int(w.gen_cnt[1].item())]
        if short:
            raise RuntimeError("forward(n_tgt_tokens=) was smaller than the number of non-pad targets in a batch of shape(s) %s: "
                               "the loss of those steps is wrong" % short)

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

    def _sumsq_entry(self, plan, first_name, end_name, slot, skip_rows=False):
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
                    if not skip_rows:           # (skip_rows: the table's flagged rows were normed earlier, right behind their last gradient)
                        plan.append((None, (k, 3 + k), "SUMSQ_ROWS", None, self._sid))
                    if t["off"] > lo:
                        plan.append((None, (lo, t["off"], slot), "SUMSQ", None, self._sid))
                    if hi - t["end"] >= SEG_ALIGN:      # (less: only the segment's alignment padding follows)
                        plan.append((None, (t["end"], hi, slot + 5), "SUMSQ", None, self._sid))
                    return
        plan.append((None, (lo, hi, slot), "SUMSQ", None, self._sid))

    def finish_allreduce(self):
        """make the current stream wait for every gradient collective of the backward plan (call before optim_step)"""
        if self.dp_on() and self.dp._comm is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.dp._comm)

    def _run(self, plan, events=None):
        main = torch.cuda.current_stream(self.dev)
        side = self.side_stream if self.use_side_stream else main
        aux = self.aux_stream if (self.use_side_stream and self.use_aux_stream) else side
        uses_tgt = self._plan_tgt.get(id(plan))
        if uses_tgt is None or uses_tgt[0] != len(plan):
            uses_tgt = self._plan_tgt[id(plan)] = (len(plan), any(en[4] == 3 for en in plan))
        uses_tgt = uses_tgt[1]
        if self._pending_bg is not None and not any(en[2] == "BG_FLUSH" for en in plan):
            self.wait_background()  # (a plan that does not place the held-back half of the optimiser step itself: issue it first, all of it)
        tgt = (self.tgt_stream if aux is self.aux_stream else aux) if uses_tgt else aux
        ts = (main, side, aux, tgt)
        hs = (main.cuda_stream, side.cuda_stream, aux.cuda_stream, tgt.cuda_stream)
        trace, last = self.trace, None
        if trace is None:
            # (the host's share of a step is ~100 of these iterations: kernel launches take the short way, everything else _exec)
            single = side is main
            for entry in plan:
                fn = entry[0]
                if fn is not None:
                    rc = fn(*entry[1], hs[entry[4]])
                    if rc != 0:
                        L.check(rc, entry[2])
                else:
                    self._exec(entry, ts, hs, events, single)
            return
        for entry in plan:
            fn, args, name, _keep, sid = entry
            if trace is not None and sid == 0 and name != last and (self.trace_only is None or name in self.trace_only):     # tools/phase_times.py: timing events at phase changes
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(main)
                trace.append((name, ev))
                last = name
            self._exec(entry, ts, hs, events, side is main)

    def _exec(self, entry, ts, hs, events, single_stream):
        fn, args, name, _keep, sid = entry
        if fn is None:
            if name == "SUMSQ":
                if self.dp_on() and self.dp.sharded:
                    return          # sharded optimiser: every rank takes the norm of ITS shards behind the segment's reduce-scatter
                lo, hi, slot = args
                if self.dp_on():    # replicated data-parallel update: behind the segments' all-reduces (COMM stream)
                    ts[sid].wait_stream(self.dp.comm_stream())
                L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * lo, hi - lo, self._sumsq.data_ptr(), slot, hs[sid]), "vmmt_sumsq")
                self._sumsq_by_plan = True
                return
            if name == "SUMSQ_ROWS":
                k, slot = args
                t = self.row_tables[k]
                L.check(self.lib.vmmt_sumsq_rows(self.flat_g.data_ptr() + 4 * t["off"], t["R"], t["C"], t["flags"].data_ptr(), t["hist"].data_ptr(),
                                                 t["rowsq"].data_ptr(), self._sumsq.data_ptr(), slot, hs[sid]), "vmmt_sumsq_rows")
                return
            if name == "KL_ALLREDUCE":
                # (only the free-bits test reads the KL sum -- latent_bwd_kernel: without it the latent backward needs nothing from the other
                #  ranks and the aux stream's chain starts without a collective in front of it)
                if self.dp_on() and self._kl_sum_needed:
                    # one float, on the COMM stream like the gradient segments (one communicator: one collective at a time), behind the
                    # background half of the last optimiser step, whose parameter all-gathers run on the side stream
                    ws = self._cur_ws
                    with torch.cuda.stream(ts[sid]):
                        ws.kl_global.copy_(ws.stats[L.STAT_KL_SUM:L.STAT_KL_SUM + 1])
                    comm = self.dp.comm_stream()
                    ev = self.global_events.get("opt_side_done")
                    if ev is not None:
                        comm.wait_event(ev)
                    self.dp.on_comm(ts[sid], lambda: self.dp.all_reduce_tensor(ws.kl_global))
                    ts[sid].wait_stream(comm)
                return
            if name == "ALLREDUCE":
                if self.dp_on():
                    lo, hi = args
                    comm = self.dp.reduce_segment(self.flat_g, lo, hi, ts[sid])
                    if self.dp.sharded and (lo, hi) in self.segments:
                        # the norm of this rank's shard of the segment right behind the collective, on the same stream: only the
                        # LAST segment's norm is left for the step's tail (optim_step).  Slots were cleared by the forward plan
                        a, b = self.dp.shard(lo, hi)
                        if b > a:
                            L.check(self.lib.vmmt_sumsq(self.flat_g.data_ptr() + 4 * a, b - a, self._sumsq.data_ptr(),
                                                        self.segments.index((lo, hi)), comm.cuda_stream), "vmmt_sumsq")
                        self._normed.add(self.segments.index((lo, hi)))
                return
            if name == "BG_FLUSH":
                self._flush_bg(ts[0], parts=1)
                return
            if name == "BG_FLUSH2":
                self._flush_bg(stream=None if single_stream else ts[sid])
                return
            if single_stream:
                return
            if name == "EV_RECORD":
                ev = events.get(args)
                if ev is None:
                    ev = events[args] = torch.cuda.Event()
                ev.record(ts[sid])
            else:
                if args == "opt_side_done":
                    self._flush_bg()
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
        self._plan_tgt.clear()              # (keyed by id(plan): a freed plan's id may be handed to a new one)
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
