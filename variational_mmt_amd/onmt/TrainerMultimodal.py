"""Mirror of onmt.TrainerMultimodal / VIStatistics (reference: onmt/TrainerMultimodal.py:32-718).

Differences that do not change results: the image-feature arrays are uploaded to HBM once and rows are gathered on the
device from `batch.indices` (reference: host fancy-index + H2D per step, :632-639); statistics stay on the device and
are read back only when they are printed or queried (reference: `.data.clone()` syncs every step)."""
import math
import os
import pickle
import shutil
import sys
import tempfile
import time

import torch
import torch.nn as nn

from . import io
from .Trainer import Statistics  # noqa: F401
from .Utils import MODEL_TYPES

_FIELDS = ("nmt_loss", "elbo_loss", "td_kl_before", "td_kl_after", "image_feats_loss", "image_feats_cos", "n_words", "n_correct")


class VIStatistics(object):
    """Accumulator with the reference's fields; values produced by the kernels are kept as device vectors (`pending`)
    until a field is read."""

    def __init__(self, multimodal_model_type, loss_data=None, n_words=0, n_correct=0, pending=None):
        assert multimodal_model_type in MODEL_TYPES
        self.multimodal_model_type = multimodal_model_type
        self.progress_state_train, self.progress_state_valid = [], []
        self.two_step_image_prediction = False
        self.image_loss_type = "logprob"
        self._v = dict.fromkeys(_FIELDS, 0.0)
        self._v["n_words"], self._v["n_correct"] = n_words, n_correct
        self.image_pixels_loss = self.image_pixels_acc = self.img_pixels_acc = 0.0
        self.te_kl = self.td_kl = 0.0
        self.td_kl_multiplier = 1.0
        if loss_data is not None:
            self._v.update(nmt_loss=loss_data["nmt"], elbo_loss=loss_data["elbo"], td_kl_before=loss_data["td_kl_before"],
                           td_kl_after=loss_data["td_kl_after"], image_feats_loss=loss_data.get("img_feats_loss", 0.0),
                           image_feats_cos=loss_data.get("img_feats_cos", 0.0))
            self.td_kl_multiplier = loss_data.get("td_kl_multiplier", 1.0)
        self._pending = [pending] if pending is not None else []
        if pending is not None:
            self.td_kl_multiplier = pending[3]
        self.n_src_words = 0
        self.n_updates = 0
        self.start_time = time.time()

    def _materialise(self):
        if not self._pending:
            return
        from .. import _lib as L
        vals = torch.stack([p[0] for p in self._pending]).tolist()          # one D2H copy
        for s, pend in zip(vals, self._pending):
            _t, B, Bg, mult, fb, margin = pend[:6]
            has_global = len(pend) > 6 and pend[6]
            # single process: Bg == B.  Data parallel: this rank's SHARE of the global figures (shares add up over the ranks to
            # the statistics of the single-process run on the whole minibatch); slot STAT_COUNT-1 carries the all-reduced KL sum
            kl_before = s[L.STAT_KL_SUM] / Bg
            kl_after = kl_before * mult
            if fb:
                kl_glob = (s[L.STAT_COUNT - 1] if has_global else s[L.STAT_KL_SUM]) / Bg * mult
                if kl_glob < margin:
                    kl_after = margin * B / Bg
            v = self._v
            v["nmt_loss"] += s[L.STAT_NLL]
            v["td_kl_before"] += kl_before
            v["td_kl_after"] += kl_after
            v["image_feats_loss"] += s[L.STAT_IMG_LOGPROB]
            v["image_feats_cos"] += s[L.STAT_IMG_COS] / B
            v["elbo_loss"] += s[L.STAT_NLL] - s[L.STAT_IMG_LOGPROB] + kl_after
            v["n_words"] += int(round(s[L.STAT_NWORDS]))
            v["n_correct"] += int(round(s[L.STAT_NCORRECT]))
        self._pending = []

    def __getattr__(self, k):
        if k in _FIELDS:
            self._materialise()
            return self._v[k]
        raise AttributeError(k)

    def update(self, stat):
        if isinstance(stat, VIStatistics):
            self._pending.extend(stat._pending)
            for k in _FIELDS:
                self._v[k] += stat._v[k]
            self.td_kl_multiplier = stat.td_kl_multiplier
        self.n_updates += 1

    def accuracy(self):
        return 100 * (self.n_correct / self.n_words)

    def ppl(self):
        return math.exp(min(self.nmt_loss / self.n_words, 100))

    def elapsed_time(self):
        return time.time() - self.start_time

    def output(self, epoch, batch, n_batches, start):
        t = self.elapsed_time()
        n = max(self.n_updates, 1)
        print(("Epoch %2d, %5d/%5d; acc: %6.2f; ppl: %6.2f; td-kl-before (avg.): %6.2f; td-kl-after (avg.): %6.2f; "
               "td-kl-multiplier: %2.2f;img-feats-loss (avg.): %6.2f; img-feats-cos (avg.): %6.2f; elbo (avg.): %6.2f; "
               "%3.0f src tok/s; %3.0f tgt tok/s; %6.0f s elapsed") %
              (epoch, batch, n_batches, self.accuracy(), self.ppl(), self.td_kl_before / n, self.td_kl_after / n,
               self.td_kl_multiplier, self.image_feats_loss / n, self.image_feats_cos / n, self.elbo_loss / n,
               self.n_src_words / (t + 1e-5), self.n_words / (t + 1e-5), time.time() - start))
        self.n_updates = 0
        sys.stdout.flush()

    def log(self, prefix, experiment, lr):
        """scalars to a crayon experiment (`-exp_host`; TrainerMultimodal.py:186-201): anything with add_scalar_value(name, value)"""
        vals = [("ppl", self.ppl()), ("accuracy", self.accuracy()), ("img-feats-loss", self.image_feats_loss),
                ("img-feats-cos", self.image_feats_cos), ("img-pixels-loss", self.image_pixels_loss), ("img-pixels-acc", self.img_pixels_acc),
                ("td-kl-before", self.td_kl_before), ("td-kl-after", self.td_kl_after), ("td-kl", self.td_kl),
                ("td-kl-multiplier", self.td_kl_multiplier), ("elbo", self.elbo_loss),
                ("tgtper", self.n_words / (self.elapsed_time() + 1e-5)), ("lr", lr)]
        for name, v in vals:
            experiment.add_scalar_value(prefix + "_" + name, float(v))

    def save_progress(self, lr, model_updates, epoch, split):
        """The reference appends a dict per update (unbounded, with a device sync each time); here one entry per call
        without forcing a read-back of pending device statistics."""
        assert split in ("train", "valid")
        rec = {"epoch": epoch, "model_updates": model_updates, "elapsed_time": self.elapsed_time(), "lr": lr}
        (self.progress_state_train if split == "train" else self.progress_state_valid).append(rec)


class _PickledAs(object):
    """pickles as an instance of `cls` whose state is `state` (taken on the training thread: pickling it touches no device)"""

    def __init__(self, cls, state):
        self._cls, self._state = cls, state

    def __reduce__(self):
        import copyreg
        return (copyreg._reconstructor, (self._cls, object, None), self._state)


class _CheckpointWriter(object):
    """`torch.save` of finished host snapshots on a background thread, one file after the other in the order they were handed over.
    An epoch of the run scripts' recipe trains in ~1.4 s; serialising + writing the ~0.7 GB checkpoint took 0.36 s of every one
    (tools/driver_workflow.py).  `wait()` returns when everything handed over is on disk and re-raises a writer's exception."""

    def __init__(self):
        import queue
        import threading
        self.q = queue.Queue()
        self.err = None
        self.t = threading.Thread(target=self._run, name="vmmt-checkpoint-writer", daemon=True)
        self.t.start()
        import atexit
        atexit.register(self.wait)

    def _run(self):
        while True:
            job = self.q.get()
            try:
                if job is not None and self.err is None:     # (after a failure nothing more is written; put() raises it at the next hand-over)
                    obj, fname = job
                    tmp = fname + ".partial"
                    torch.save(obj, tmp)
                    os.replace(tmp, fname)          # readers never see a half-written file
            except BaseException as ex:             # noqa: B902  (reported by wait())
                self.err = ex
            finally:
                self.q.task_done()

    def put(self, obj, fname):
        """hand a snapshot over; a failure of an EARLIER write (disk full, bad path) is raised here, at the next checkpoint, instead of
        hours later at the end of the run while every checkpoint in between was silently dropped"""
        if self.err is not None:
            err, self.err = self.err, None
            raise err
        self.q.put((obj, fname))

    def wait(self):
        self.q.join()
        if self.err is not None:
            err, self.err = self.err, None
            raise err


class _NoEarlyStop(object):
    """stand-in when no model options are given (tests, benchmarks): perplexity criterion, nothing to evaluate"""

    def __init__(self, criteria, every):
        self.early_stop_criteria = criteria
        self.evaluate_every_nupdates = every
        self.signal_early_stopping = False
        self.results_bleu, self.results_meteor = {}, {}


class TrainerMultimodal(object):
    def __init__(self, model, train_loss, valid_loss, optim, trunc_size=0, shard_size=32, data_type="text",
                 norm_method="sents", grad_accum_count=1, train_img_feats=None, valid_img_feats=None, train_img_vecs=None,
                 valid_img_vecs=None, multimodal_model_type=None, model_updates=0, model_opt=None, fields=None):
        self.model, self.train_loss, self.valid_loss, self.optim = model, train_loss, valid_loss, optim
        self.trunc_size, self.shard_size, self.data_type, self.norm_method = trunc_size, shard_size, data_type, norm_method
        self.grad_accum_count = grad_accum_count
        self.multimodal_model_type = multimodal_model_type
        self.model_updates = model_updates
        self.model_opt, self.fields = model_opt, fields
        crit = getattr(model_opt, "early_stopping_criteria", "perplexity") if model_opt is not None else "perplexity"
        if model_opt is not None and crit not in (None, "perplexity"):
            # TrainerMultimodal.py:281-288
            from .EarlyStop import EarlyStop
            self.early_stop = EarlyStop(model_opt.src, model_opt.tgt, crit, model_opt.start_early_stopping_at,
                                        model_opt.evaluate_every_n_model_updates, model_opt.patience,
                                        multimodal_model_type=multimodal_model_type,
                                        img_fname=model_opt.path_to_valid_img_feats, gpuid=getattr(model_opt, "gpuid", 0))
            self.early_stop.attach_model(model, fields)
        else:
            self.early_stop = _NoEarlyStop(crit, getattr(model_opt, "evaluate_every_n_model_updates", 500) if model_opt else 500)
        self.n_model_updates = 0
        self._epoch = 0
        # drop_checkpoint returns with the file on disk, as the reference's does -- unless the caller opts into the background writer
        # and calls finish_checkpoints() before it reads a file back (the build's own driver does: train_mm_vi_model1.py)
        self.async_checkpoints = False
        self._writer = None
        assert train_img_feats is not None and valid_img_feats is not None, "Must provide training/validation image features!"
        assert multimodal_model_type in (None, "vi-model1")
        assert grad_accum_count == 1, "gradient accumulation is not on the hot path (reference default 1)"
        assert trunc_size == 0, "truncated BPTT is not on the hot path (reference recipes use 0)"
        self.train_img_feats, self.valid_img_feats = train_img_feats, valid_img_feats
        self.model.set_image_tables(train=train_img_feats, valid=valid_img_feats)
        from .. import dp
        self.dp = dp.GradSync(self.model.engine)          # no-op unless torch.distributed is initialised with > 1 rank
        self.dp_resync_every = 2000                       # replicas are bit-identical by construction; cheap insurance
        self.model.train()

    # ------------------------------------------------------------------------------------------------------------
    def _prep(self, batch):
        src = io.make_features(batch, "src", self.data_type)
        _, src_lengths = batch.src
        tgt = io.make_features(batch, "tgt")
        _, tgt_lengths = batch.tgt
        batch.tgt = batch.tgt[0]          # the reference mutates the batch the same way (TrainerMultimodal.py:677)
        return src, src_lengths, tgt, tgt_lengths

    def train(self, train_iter, epoch, report_func=None):
        total_stats = VIStatistics(self.multimodal_model_type)
        report_stats = VIStatistics(self.multimodal_model_type)
        try:
            num_batches = len(train_iter)
        except (NotImplementedError, TypeError):
            num_batches = -1
        return self._train_loop(train_iter, epoch, report_func, total_stats, report_stats, num_batches)

    def _train_loop(self, train_iter, epoch, report_func, total_stats, report_stats, num_batches):
        # inside this loop an update is followed by the next batch's forward: the engine may hold the side-stream half of an update back
        # until that forward's head is through (Engine.bg_after_head); whoever reads parameters inside the loop (a checkpoint, a
        # translation) goes through state_dict() / the launch plans, which issue it first.  The model's nn.Parameter objects ALIAS the arena
        # (Models.py): code that reads them directly inside this loop (Optim.params, a hook) must call engine.wait_background() first, or
        # it sees the decoder half one update behind
        eng = self.model.engine
        eng.hold_back = True
        try:
            return self._train_batches(train_iter, epoch, report_func, total_stats, report_stats, num_batches)
        finally:
            eng.hold_back = False
            eng.wait_background()

    def _train_batches(self, train_iter, epoch, report_func, total_stats, report_stats, num_batches):
        for idx, batch in enumerate(train_iter):
            if hasattr(train_iter, "get_cur_dataset"):
                self.train_loss.cur_dataset = train_iter.get_cur_dataset()
            # (under data parallelism these are this rank's figures; _gradient_accumulation turns them into the global ones)
            if self.norm_method == "tokens":
                normalization = getattr(batch, "global_ntokens", None)
                if normalization is None:
                    normalization = int(batch.tgt[0][1:].ne(self.train_loss.padding_idx).sum())
            else:
                normalization = getattr(batch, "global_batch_size", batch.batch_size)
            self._gradient_accumulation([batch], total_stats, report_stats, normalization)
            if report_func is not None:
                report_stats = report_func(epoch, idx, num_batches, total_stats.start_time, self.optim.lr, report_stats,
                                           self.multimodal_model_type)
            self.n_model_updates += 1
            # BLEU / METEOR model selection and early stopping (TrainerMultimodal.py:372-396)
            if self.early_stop.early_stop_criteria not in (None, "perplexity") and \
                    self.n_model_updates % self.early_stop.evaluate_every_nupdates == 0:
                # (the reference writes a temporary checkpoint first because its translation runs in a subprocess on that file, then
                #  moves or deletes it, :372-396; here the live model translates, so the checkpoint is written only when it is the best)
                if self.early_stop.add_run(None, self.n_model_updates):
                    self.drop_metric_scores(self.model_opt, epoch, self.fields, valid_stats=None, overwrite=True,
                                            checkpoint_type="best")
                    self.drop_checkpoint(self.model_opt, epoch, self.fields, valid_stats=None, overwrite=True, checkpoint_type="best")
                if self.early_stop.signal_early_stopping:
                    break
        self.model.engine.check_async_errors()
        return total_stats

    def validate(self, valid_iter):
        self.model.eval()
        stats = VIStatistics(self.multimodal_model_type)
        for batch in valid_iter:
            if hasattr(valid_iter, "get_cur_dataset"):
                self.valid_loss.cur_dataset = valid_iter.get_cur_dataset()
            src, src_lengths, tgt, tgt_lengths = self._prep(batch)
            outputs, attns, _ = self.model(src, tgt, src_lengths, tgt_lengths, None, img_indices=batch.indices,
                                           img_table=self.model._tables["valid"], padding_token=self.train_loss.padding_idx)
            stats.update(self.valid_loss.monolithic_compute_loss(batch, outputs, attns))
        stats.save_progress(self.optim.lr, self.model_updates, self._epoch, "valid")
        self.model.train()
        return stats

    def epoch_step(self, ppl, epoch):
        return self.optim.update_learning_rate(ppl, epoch)

    def _gradient_accumulation(self, true_batchs, total_stats, report_stats, normalization):
        for batch in true_batchs:
            src, src_lengths, tgt, tgt_lengths = self._prep(batch)
            report_stats.n_src_words += int(src.shape[0] * src.shape[1])   # upper bound without a device sync
            world = self.dp.world
            norm = normalization
            if world == 1:
                self.train_loss.batch_global = batch.batch_size
            elif hasattr(batch, "global_batch_size"):
                # the sharding iterator (onmt.io.OrderedIterator(dp_rank=, dp_world=)) cut a global minibatch on the host:
                # the global batch size / token count are known here without any collective or host sync
                self.train_loss.batch_global = batch.global_batch_size
            else:
                # a per-rank loader: ONE blocking all-reduce for both figures (slow path, kept for compatibility)
                self.train_loss.batch_global, norm = self.dp.global_sizes(batch.batch_size, normalization)
            outputs, attns, _ = self.model(src, tgt, src_lengths, tgt_lengths, None, img_indices=batch.indices,
                                           img_table=self.model._tables["train"], padding_token=self.train_loss.padding_idx,
                                           n_tgt_tokens=getattr(batch, "n_tgt_tokens", None))
            batch_stats = self.train_loss.sharded_compute_loss(batch, outputs, attns, 0, tgt.shape[0], self.shard_size, norm)
            self.model_updates += 1
            self.dp.all_reduce()
            self.optim.step()
            if self.dp.active() and self.dp_resync_every and self.model_updates % self.dp_resync_every == 0:
                self.dp.broadcast_replica(0)
            total_stats.update(batch_stats)
            report_stats.update(batch_stats)

    def drop_checkpoint(self, opt, epoch, fields, valid_stats, train_stats=None, overwrite=False, checkpoint_type="last",
                        temporary=False):
        """Same checkpoint dict as the reference (TrainerMultimodal.py:554-622): keys model / generator / vocab / opt /
        epoch / optim; parameter names per SURVEY.md Appendix B.
        Data parallelism: EVERY rank calls this at the same point of the run -- collecting the sharded optimiser's moments is a
        collective (dp.gather_moments) -- and rank 0 alone snapshots and writes the file (the other ranks return its name)."""
        assert checkpoint_type in ("last", "best")
        self.model.engine.check_async_errors()         # never write parameters that a timed-out device hand-off may have corrupted
        self.dp.gather_moments()                       # sharded data-parallel optimiser: collect Adam's moments from their owners
        if not overwrite:
            fname = "%s_acc_%.2f_ppl_%.2f_e%d.pt" % (opt.save_model, valid_stats.accuracy(), valid_stats.ppl(), epoch)
        elif checkpoint_type == "best":
            crit = self.early_stop.early_stop_criteria
            if crit not in ("bleu", "meteor"):
                raise Exception("Metric not supported.")
            fname = "%s_BestModel%s.pt" % (opt.save_model, crit.capitalize())
        else:
            fname = "%s_MostCurrentModel.pt" % opt.save_model
        if self.dp.rank != 0:
            return (None, fname) if temporary else fname
        # (state_dict() clones on the device, .cpu() makes a new host tensor of each; an entry that already lives on the host would be
        #  aliased by .cpu() while training continues under the background writer, hence the explicit copy)
        sd = {k: (v.detach().cpu() if v.is_cuda else v.detach().clone()) for k, v in self.model.state_dict().items()}
        model_sd = {k: v for k, v in sd.items() if "generator" not in k}
        gen_sd = {k[len("generator."):]: v for k, v in sd.items() if k.startswith("generator.")}
        self.optim._ckpt_cpu = sd          # Optim.__getstate__ wraps these same tensors: one copy of each in the file
        # the optimiser's state (learning-rate schedule + Adam's moments, read from the device) is taken HERE, on the training thread;
        # what goes to the writer is host memory only and pickles as the same `onmt.Optim.Optim`
        frozen = _PickledAs(type(self.optim), self.optim.__getstate__())
        self.optim._ckpt_cpu = None
        checkpoint = {"model": model_sd, "generator": gen_sd, "vocab": io.save_fields_to_vocab(fields), "opt": opt,
                      "epoch": epoch, "optim": frozen}
        if temporary:
            tf = tempfile.NamedTemporaryFile(delete=False)
            tf.close()
            torch.save(checkpoint, tf.name)
            return tf.name, fname
        if self.async_checkpoints:
            if self._writer is None:
                self._writer = _CheckpointWriter()
            self._writer.put(checkpoint, fname)
        else:
            torch.save(checkpoint, fname)
        return fname

    def finish_checkpoints(self):
        """every checkpoint handed to the background writer is on disk when this returns (the drivers call it before they read a
        checkpoint back and at the end of the run; an interpreter exit waits as well)"""
        if self._writer is not None:
            self._writer.wait()


    def drop_metric_scores(self, opt, epoch, fields, valid_stats, overwrite=False, checkpoint_type="last"):
        """TrainerMultimodal.py:491-551: pickle of the metric scores next to the checkpoint -- for 'best' the best BLEU / METEOR
        and the number of model updates that produced it (`<save_model>_BestModel<Metric>.pkl`), for 'last' every score so far
        (`<save_model>_MostCurrentModel.pkl`)."""
        assert checkpoint_type in ("last", "best")
        es = self.early_stop
        if getattr(getattr(self, "dp", None), "rank", 0) != 0:      # data parallelism: one writer (every rank holds the same scores)
            return "%s_%s.pkl" % (opt.save_model, "MostCurrentModel" if checkpoint_type == "last" else "BestModel" + str(es.early_stop_criteria).capitalize())
        rec = {}
        if checkpoint_type == "best":
            if es.early_stop_criteria == "bleu":
                metric, table, other = "bleu", es.results_bleu, es.results_meteor
            elif es.early_stop_criteria == "meteor":
                metric, table, other = "meteor", es.results_meteor, es.results_bleu
            else:
                raise Exception("Metric not supported.")
            n_best, best = sorted(table.items(), key=lambda kv: (kv[1], kv[0]))[-1]
            rec["n_updates"] = n_best
            rec[metric] = float(best)
            rec["meteor" if metric == "bleu" else "bleu"] = float(other[n_best])
            fname = "%s_BestModel%s.pkl" % (opt.save_model, metric.capitalize())
        else:
            fname = "%s_MostCurrentModel.pkl" % opt.save_model
            rec["n_updates"] = self.n_model_updates
            rec["bleu"] = list(es.results_bleu.values())
            rec["meteor"] = list(es.results_meteor.values())
        with open(fname, "wb") as f:
            pickle.dump(rec, f, pickle.HIGHEST_PROTOCOL)
        return fname
