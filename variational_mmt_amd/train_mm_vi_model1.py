"""The training driver of the MI355X build: `python -m variational_mmt_amd.train_mm_vi_model1 <flags>` (or the launcher
`train_mm_vi_model1.py` at the repository root) takes the command lines of the reference's `train_mm_vi_model1.py` -- the ones in
run_translated_m30k_only.sh:46-71 and run_additional_data.sh:72-144 run unchanged -- and performs the same sequence on the
mirrored `onmt` surface:

    flags (opts.py)  ->  image features HDF5 -> HBM  ->  checkpoint? (-train_from [-finetune])  ->  dataset + vocabulary pickles
    ->  make_vi_model_mmt  ->  Optim  ->  TrainerMultimodal: [validate] / train / validate / lr schedule / checkpoints per epoch,
    BLEU / METEOR model selection and early stopping inside the epoch

Reference: train_mm_vi_model1.py:28-90 (flag handling), :246-335 (epoch loop), :347-454 (datasets, fields, model, optimiser),
:457-581 (main).  What differs on purpose:
  * the image-feature files are streamed straight into HBM (features.load_image_table; standardisation on the device) instead of
    being read into host numpy arrays that are indexed and copied every step;
  * `-gpuid` names the GPU of THIS process; under `torch.distributed` (one process per GPU, RANK / WORLD_SIZE in the environment)
    the training iterator shards every minibatch over the ranks and only rank 0 writes checkpoints (the reference stops at
    `len(gpuid) > 1`, :73-75);
  * no global `torch.set_default_tensor_type`, no crayon logging.
Nothing here computes on the GPU itself: every number comes from libvmmt.so through the `onmt` mirror.
"""
import argparse
import glob
import os
import random
import sys

import torch

from . import install_as_onmt, opts


def parse(argv=None):
    ap = argparse.ArgumentParser(description="train_mm_vi_model1.py", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    opts.add_md_help_argument(ap)
    opts.model_opts(ap)
    opts.train_opts(ap)
    opts.train_mm_vi_model1_opts(ap)
    opt = opts.finalise(ap.parse_args(argv))
    if opt.early_stopping_criteria != "perplexity" and not (opt.src and opt.tgt):
        ap.error("-early_stopping_criteria %s needs the validation -src and -tgt files" % opt.early_stopping_criteria)
    for flag in ("path_to_train_img_feats", "path_to_valid_img_feats"):
        if not os.path.isfile(getattr(opt, flag)):
            ap.error("-%s: no such file: %s" % (flag, getattr(opt, flag)))
    if opt.two_step_image_prediction and not opt.use_local_image_features:
        ap.error("-two_step_image_prediction needs --use_local_image_features")
    if len(opt.gpuid) > 1:
        ap.error("-gpuid takes ONE id per process; for several GPUs start one process per GPU (torch.distributed.run)")
    if not opt.gpuid:
        ap.error("the MI355X build has no CPU path: pass -gpuid 0")
    return opt


class _Shards(object):
    """the `<data>.<split>.N.pt` files (or the single `<data>.<split>.pt`), opened one at a time, as one stream of minibatches
    (reference: lazily_load_dataset :347-385 + DatasetLazyIter :129-187)"""

    def __init__(self, onmt, opt, split, fields, rank, world):
        self.onmt, self.opt, self.split, self.fields = onmt, opt, split, fields
        self.rank, self.world = rank, world
        numbered = glob.glob("%s.%s.[0-9]*.pt" % (opt.data, split))
        self.files = sorted(numbered) or ["%s.%s.pt" % (opt.data, split)]
        self.cur_dataset = None
        self._len = None

    def get_cur_dataset(self):
        return self.cur_dataset

    _resident = {}          # path -> (mtime, dataset): a split that is ONE file stays loaded between epochs

    def _load(self, path):
        """lazily_load_dataset re-reads every shard in every epoch because a corpus in many shards does not fit in memory at once
        (:347-385); a split in a single file (Multi30k: 29 000 pairs) is kept -- unpickling it and numericalising its words again
        cost 0.3 s of a 1.7-s epoch (tools/driver_workflow.py)"""
        if len(self.files) != 1:
            return self.onmt.io.load_dataset(path)
        mtime = os.path.getmtime(path)
        hit = _Shards._resident.get(path)
        if hit is None or hit[0] != mtime:
            hit = _Shards._resident[path] = (mtime, self.onmt.io.load_dataset(path))
        return hit[1]

    def _iterator(self, path):
        ds = self._load(path)
        print("Loading %s dataset from %s, number of examples: %d" % (self.split, path, len(ds)))
        ds.fields = self.fields
        self.cur_dataset = ds
        train = self.split == "train"
        budget = None
        if train and self.opt.batch_type == "tokens":
            budget = lambda new, count, sofar: sofar + max(len(new.tgt), len(new.src)) + 1        # noqa: E731
        # validation is not sharded: every rank scores the whole set (statistics are then identical everywhere)
        dp = dict(dp_rank=self.rank, dp_world=self.world) if (train and self.world > 1) else {}
        return self.onmt.io.OrderedIterator(dataset=ds, batch_size=self.opt.batch_size if train else self.opt.valid_batch_size,
                                            batch_size_fn=budget, device=self.opt.gpuid[0], sort=False, train=train,
                                            sort_within_batch=True, repeat=False, **dp)

    def __iter__(self):
        for path in self.files:
            it = self._iterator(path)
            self._len = len(it)
            for b in it:
                yield b

    def __len__(self):
        if self._len is None:            # like the reference: the batch count of the shard at hand
            self._len = len(self._iterator(self.files[0]))
        return self._len


def _progress(opt, onmt):
    def report(epoch, batch, num_batches, start_time, lr, stats, mm_type):
        if (batch + 1) % opt.report_every == 0:
            stats.output(epoch, batch + 1, num_batches, start_time)
            stats = onmt.VIStatistics(mm_type)
        return stats
    return report


def _loss(onmt, model, vocab, opt, training):
    kw = dict(label_smoothing=opt.label_smoothing, kl_annealing_current=opt.kl_annealing_start,
              kl_annealing_increment=opt.kl_annealing_increment, kl_annealing_warmup_steps=opt.kl_annealing_warmup_steps,
              image_loss_type=opt.image_loss, use_local_image_features=opt.use_local_image_features,
              two_step_image_prediction=opt.two_step_image_prediction)
    if training:         # validation always weighs the KL term fully (train_mm_vi_model1.py:228-241)
        kw.update(use_kl_annealing=opt.use_kl_annealing, use_kl_freebits=opt.use_kl_freebits, kl_freebits_margin=opt.kl_freebits_margin)
    return onmt.VILoss.NMTVIModel1LossCompute(model.generator, vocab, **kw).cuda()


def _image_tables(opt, device):
    from .features import load_image_table
    if opt.image_loss == "categorical":
        raise NotImplementedError("--image_loss categorical (image pixels) is outside the VI_Model1 hot path")
    node = "local_feats" if opt.use_local_image_features else "global_feats" if opt.use_global_image_features else "logits"
    print("Using %s image features..." % {"local_feats": "local", "global_feats": "global", "logits": "posterior class"}[node])
    mean = std = None
    if opt.use_standardised_image_features:
        if node != "global_feats":
            raise ValueError("--use_standardised_image_features goes with --use_global_image_features only")
        mean, std = opt.path_to_mean_train_img_feats, opt.path_to_std_train_img_feats
        if not (mean and std and os.path.isfile(mean) and os.path.isfile(std)):
            raise ValueError("Problem loading training set's mean and variance from `-path_to_mean_train_img_feats` and "
                             "`-path_to_std_train_img_feats`.")
    return [load_image_table(p, node, device=device, mean_path=mean, std_path=std)
            for p in (opt.path_to_train_img_feats, opt.path_to_valid_img_feats)]


def main(argv=None):
    opt = parse(argv)
    onmt = install_as_onmt()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if opt.seed > 0:
        random.seed(opt.seed)
        torch.manual_seed(opt.seed)
    torch.cuda.set_device(opt.gpuid[0])
    device = torch.device("cuda", opt.gpuid[0])
    print("Using GPU")
    if world > 1 and not torch.distributed.is_initialized():
        # RCCL ("nccl" on ROCm) over xGMI, one process per GPU.  VMMT_DP_BACKEND=gloo: rehearsal of the multi-rank driver with every
        # rank on ONE GPU (RCCL refuses two ranks on one device), tests/test_gpu_driver_cli.py
        backend = os.environ.get("VMMT_DP_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)
    train_feats, valid_feats = _image_tables(opt, device)

    # ---- where to start from: -train_from continues a run (its options, its optimiser), with -finetune only the weights are kept
    checkpoint, model_opt = None, opt
    if opt.train_from:
        print("Loading checkpoint from %s" % opt.train_from)
        checkpoint = torch.load(opt.train_from, map_location="cpu", weights_only=False)
        if not opt.finetune:
            model_opt = checkpoint["opt"]
        opt.start_epoch = checkpoint["epoch"] + 1

    # ---- vocabularies -> fields (those the examples actually carry)
    first = _Shards(onmt, opt, "train", None, rank, world).files[0]
    probe = onmt.io.load_dataset(first)
    data_type = probe.data_type
    if checkpoint is not None:
        print("Loading vocab from checkpoint at %s." % opt.train_from)
        vocab = checkpoint["vocab"]
    else:
        vocab = onmt.io.load_vocab(opt.data + ".vocab.pt")
    fields = onmt.io.load_fields_from_vocab(vocab, data_type)
    fields = dict((k, f) for k, f in fields.items() if k in probe.examples[0].__dict__)
    del probe
    print(" * vocabulary size. source = %d; target = %d" % (len(fields["src"].vocab), len(fields["tgt"].vocab)))
    for side in ("src", "tgt"):
        for j, name in enumerate(onmt.io.collect_features(fields, side=side)):
            print(" * %s feature %d size = %d" % (side, j, len(fields[name].vocab)))

    # ---- model, optimiser
    print("Building model...")
    model = onmt.ModelConstructor.make_vi_model_mmt(model_opt, fields, True, checkpoint)
    n_enc = sum(p.nelement() for n, p in model.named_parameters() if "encoder" in n)
    n_all = sum(p.nelement() for p in model.parameters())
    print("* number of parameters: %d\nencoder:  %d\ndecoder:  %d" % (n_all, n_enc, n_all - n_enc))
    os.makedirs(os.path.dirname(os.path.abspath(opt.save_model)), exist_ok=True)
    if checkpoint is not None and not opt.finetune:
        print("Loading optimizer from checkpoint.")
        optim = checkpoint["optim"]
        optim.optimizer.load_state_dict(checkpoint["optim"].optimizer.state_dict())
    else:
        print("Making optimizer for training.")
        optim = onmt.Optim(opt.optim, opt.learning_rate, opt.max_grad_norm, lr_decay=opt.learning_rate_decay,
                           start_decay_at=opt.start_decay_at, beta1=opt.adam_beta1, beta2=opt.adam_beta2,
                           adagrad_accum=opt.adagrad_accumulator_init, decay_method=opt.decay_method,
                           warmup_steps=opt.warmup_steps, model_size=opt.rnn_size)
    optim.set_parameters(model.parameters())

    # ---- trainer and the epoch loop
    tgt_vocab = fields["tgt"].vocab
    trainer = onmt.TrainerMultimodal(model, _loss(onmt, model, tgt_vocab, opt, True), _loss(onmt, model, tgt_vocab, opt, False), optim,
                                     opt.truncated_decoder, opt.max_generator_batches, data_type, opt.normalization, opt.accum_count,
                                     train_feats, valid_feats, multimodal_model_type=opt.multimodal_model_type,
                                     train_img_vecs=None, valid_img_vecs=None, model_opt=model_opt, fields=fields)
    if world > 1:
        trainer.dp.broadcast_replica(0)
    # checkpoints: snapshot on this thread, serialise + write on a background thread (VMMT_ASYNC_CHECKPOINT=0: write in place)
    trainer.async_checkpoints = os.environ.get("VMMT_ASYNC_CHECKPOINT", "1") == "1"
    print("\nStart training...")
    print(" * number of epochs: %d, starting from Epoch %d" % (opt.epochs + 1 - opt.start_epoch, opt.start_epoch))
    print(" * batch size: %d" % opt.batch_size)
    report = _progress(opt, onmt)

    def validate():
        vs = trainer.validate(_Shards(onmt, opt, "valid", fields, rank, world))
        print("Validation perplexity: %g" % vs.ppl())
        print("Validation accuracy: %g" % vs.accuracy())
        return vs

    for epoch in range(opt.start_epoch, opt.epochs + 1):
        print("")
        if epoch == 1:
            validate()                    # the untrained model's scores, as the reference prints them
        ts = trainer.train(_Shards(onmt, opt, "train", fields, rank, world), epoch, report)
        print("Train perplexity: %g" % ts.ppl())
        print("Train accuracy: %g" % ts.accuracy())
        vs = validate()
        n = max(vs.n_updates, 1)
        print("Validation image feats nll (avg.): %g" % (vs.image_feats_loss / n))
        print("Validation image fests cosine (avg.): %g" % (vs.image_feats_cos / n))
        trainer.epoch_step(vs.ppl(), epoch)
        # every rank calls drop_checkpoint (collecting the sharded optimiser's moments is a collective); rank 0 writes the file
        if trainer.early_stop.early_stop_criteria in ("perplexity", None):
            if epoch >= opt.start_checkpoint_at:
                trainer.drop_checkpoint(model_opt, epoch, fields, vs, overwrite=opt.overwrite_model_file)
        else:    # BLEU / METEOR model selection writes the best model itself; keep the latest one for continuing the run
            trainer.drop_checkpoint(model_opt, epoch, fields, vs, overwrite=opt.overwrite_model_file, checkpoint_type="last")
            trainer.drop_metric_scores(model_opt, epoch, fields, vs, overwrite=True, checkpoint_type="last")
            print("")
        if trainer.early_stop.signal_early_stopping:
            print("WARNING: Early stopping!")
            break
    model.engine.check_async_errors()
    trainer.finish_checkpoints()          # (checkpoints are written by a background thread: on disk when main returns)
    return trainer


if __name__ == "__main__":
    main(sys.argv[1:])
