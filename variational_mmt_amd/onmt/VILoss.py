"""Mirror of NMTVIModel1LossCompute (reference: onmt/VILoss.py:59-531, onmt/Loss.py:68-132).

sharded_compute_loss == statistics + `loss.div(normalization).backward()` through the WHOLE model, executed by the HIP
backward plan (the generator sharding of the reference is an autograd memory trick; all T' rows are used: hazard H3).
KL annealing / free-bits state lives here as host scalars and advances once per call exactly like the reference
(VILoss.py:463-473, 501-511; H8)."""
from .Loss import LossComputeBase
from .TrainerMultimodal import VIStatistics


class NMTVIModel1LossCompute(LossComputeBase):
    def __init__(self, generator, tgt_vocab, normalization="sents", label_smoothing=0.0, use_kl_annealing=False,
                 use_kl_freebits=False, kl_freebits_margin=0.0, kl_annealing_current=0.0, kl_annealing_increment=0.0001,
                 kl_annealing_warmup_steps=1000, image_loss_type="logprob", use_local_image_features=False,
                 two_step_image_prediction=False):
        super(NMTVIModel1LossCompute, self).__init__(generator, tgt_vocab)
        assert label_smoothing == 0.0, "label smoothing is off in every reference recipe (opts.py:336); not on the hot path"
        assert image_loss_type == "logprob" and not use_local_image_features and not two_step_image_prediction, \
            "only the global-feature `logprob` image loss is on the hot path"
        assert self.padding_idx == 1, "kernels assume <blank> == 1 (onmt/io/DatasetBase.py)"
        self.multimodal_model_type = "vi-model1"
        self.n_model_updates = 0
        self.use_kl_annealing = use_kl_annealing
        if use_kl_annealing:
            self.kl_annealing_current = kl_annealing_current
            self.kl_annealing_increment = kl_annealing_increment
            self.kl_annealing_warmup_steps = kl_annealing_warmup_steps
        else:
            self.kl_annealing_current, self.kl_annealing_increment, self.kl_annealing_warmup_steps = 1.0, 0.0, 0
        self.use_kl_freebits = use_kl_freebits
        self.kl_freebits_margin = kl_freebits_margin if use_kl_freebits else 0.0
        self.image_loss_type = image_loss_type
        self.batch_global = None        # set by the trainer under data parallelism (sum of the ranks' batch sizes)

    def _after_loss(self):
        if self.kl_annealing_current > 1.0:
            self.kl_annealing_current = 1.0
        if self.kl_annealing_current < 1.0 and self.n_model_updates >= self.kl_annealing_warmup_steps:
            self.kl_annealing_current += self.kl_annealing_increment
        self.n_model_updates += 1

    def _stats(self, ws, mult):
        st = ws.stats.clone()
        dp = ws.e.dp
        has_global = dp is not None and dp.active() and ws.training
        if has_global:
            st[-1] = ws.kl_global[0]       # the all-reduced KL sum the backward compared with the free-bits margin (spare slot)
        return VIStatistics(self.multimodal_model_type, pending=(st, ws.B, float(self.batch_global or ws.B), mult,
                                                                 self.use_kl_freebits, self.kl_freebits_margin, has_global))

    def sharded_compute_loss(self, batch, output, attns, cur_trunc, trunc_size, shard_size, normalization):
        ws = attns["_ws"]
        mult = self.kl_annealing_current if self.use_kl_annealing else 1.0
        ws.e.loss_backward(ws, normalization=float(normalization), batch_global=float(self.batch_global or ws.B), kl_mult=mult,
                           use_freebits=self.use_kl_freebits, margin=self.kl_freebits_margin)
        st = self._stats(ws, mult)
        self._after_loss()
        return st

    def monolithic_compute_loss(self, batch, output, attns):
        ws = attns["_ws"]
        ws.e.loss(ws)
        mult = self.kl_annealing_current if self.use_kl_annealing else 1.0
        st = self._stats(ws, mult)
        self._after_loss()
        return st
