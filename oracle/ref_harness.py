"""TEST INFRASTRUCTURE -- imports the *real* reference (iacercalixto/variational_mmt,
mounted read-only at /root/reference) under torch 2.x so that golden vectors can be
generated from it.  Runs ONLY in the build container: /root/reference does not exist
on the GPU box, and nothing in the product path, `pytest -m gpu`, `smoke()` or
`bench.py` may import this module.

The reference is never copied: it is imported from where it lies.  Two third-party
packages it imports for *data loading* (torchtext 0.2.3, PyTables) are absent here; they
carry no arithmetic for the training step, so tiny import stand-ins live in
`oracle/stubs/`.  Five monkey-patches adapt torch-0.3 idioms:

  s1  os.path.isfile -> True for the hard-coded METEOR jar     (onmt/Utils.py:5,8)
  s2  `1 - bool_mask` -> `~mask`                                (onmt/modules/GlobalAttention.py:176)
  s3  torch.distributions.Normal.std -> .scale                  (onmt/modules/Dists.py:19)
  s4  torch.stack(Tensor) -> torch.stack(list(Tensor))          (onmt/Models.py:1151)
  s5  criterion output reshaped to 1 element                    (onmt/VILoss.py:478,485)
  s6  (training/sharded path only) torch.split -> clones        (onmt/VILoss.py:570-580)
  s7  (training path only) out-of-place compute_cosine          (onmt/VILoss.py:43)   [semantic "A", see H1]
  s8  (beam search only) integer tensor `/` -> floor division  (onmt/translate/Beam.py:101; torch 0.3 semantics)
"""
import argparse
import contextlib
import os
import sys
import types

import torch

REF_ROOT = "/root/reference"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
_state = {"onmt": None}


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "onmt"))


def import_reference():
    """Import the reference's `onmt` package (and `opts`) with the s1-s4 shims."""
    if _state["onmt"] is not None:
        return _state["onmt"], _state["opts"]
    assert available(), "reference not mounted"
    sys.dont_write_bytecode = True
    for p in (REF_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    assert "onmt" not in sys.modules or getattr(sys.modules["onmt"], "__file__", "").startswith(REF_ROOT), \
        "another `onmt` is already imported in this process"

    # s1
    real_isfile = os.path.isfile
    os.path.isfile = lambda p: True if str(p).endswith("meteor-1.5.jar") else real_isfile(p)
    # s2
    real_rsub = torch.Tensor.__rsub__

    def rsub(self, other):
        if self.dtype == torch.bool:
            return ~self
        return real_rsub(self, other)
    torch.Tensor.__rsub__ = rsub
    # s3
    if not hasattr(torch.distributions.Normal, "std"):
        torch.distributions.Normal.std = property(lambda s: s.scale)
    # s4
    real_stack = torch.stack

    def stack(t, *a, **kw):
        if isinstance(t, torch.Tensor):
            t = list(t)
        return real_stack(t, *a, **kw)
    torch.stack = stack

    cwd = os.getcwd()
    os.chdir(REF_ROOT)        # Utils.py:4 resolves tools/multi-bleu.perl relative to cwd
    try:
        import onmt           # noqa
        import onmt.ModelConstructor  # noqa
        import opts           # noqa
    finally:
        os.chdir(cwd)
        os.path.isfile = real_isfile
    _state["onmt"], _state["opts"] = onmt, opts
    return onmt, opts


class _V(object):
    """fake torchtext Vocab with n entries; specials follow DatasetBase.py:7-11 / IO.py:221-226."""

    def __init__(self, n, tgt):
        sp = ["<unk>", "<blank>"] + (["<s>", "</s>"] if tgt else [])
        self.itos = sp + ["w%d" % i for i in range(n - len(sp))]
        self.stoi = {w: i for i, w in enumerate(self.itos)}

    def __len__(self):
        return len(self.itos)


def make_opt(**over):
    """Build `opt` with the reference's own parsers (opts.py) + the post-processing of
    train_mm_vi_model1.py:40-48."""
    onmt, opts = import_reference()
    p = argparse.ArgumentParser()
    opts.model_opts(p)
    opts.train_opts(p)
    opts.train_mm_vi_model1_opts(p)
    argv = ["-data", "x", "--multimodal_model_type", "vi-model1", "--use_global_image_features", "--z_latent_dim", "500",
            "-path_to_train_img_feats", "resnet50.hdf5", "-path_to_valid_img_feats", "resnet50.hdf5"]
    opt = p.parse_args(argv)
    for k, v in over.items():
        assert hasattr(opt, k), k
        setattr(opt, k, v)
    if opt.word_vec_size != -1:
        opt.src_word_vec_size = opt.word_vec_size
        opt.tgt_word_vec_size = opt.word_vec_size
    if opt.layers != -1:
        opt.enc_layers = opt.layers
        opt.dec_layers = opt.layers
    opt.brnn = (opt.encoder_type == "brnn")
    return opt


def build_model(opt, vs, vt, seed=0):
    onmt, _ = import_reference()
    fields = {"src": types.SimpleNamespace(vocab=_V(vs, False)),
              "tgt": types.SimpleNamespace(vocab=_V(vt, True))}
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, False, None)
    return model, fields


def make_loss(model, fields, opt, training=True):
    """train_mm_vi_model1.py:208-247 (make_loss_compute), + shim s5."""
    onmt, _ = import_reference()
    kw = dict(label_smoothing=opt.label_smoothing,
              kl_annealing_current=opt.kl_annealing_start,
              kl_annealing_increment=opt.kl_annealing_increment,
              kl_annealing_warmup_steps=opt.kl_annealing_warmup_steps,
              image_loss_type=opt.image_loss,
              use_local_image_features=opt.use_local_image_features,
              two_step_image_prediction=opt.two_step_image_prediction)
    if training:
        kw.update(use_kl_annealing=opt.use_kl_annealing, use_kl_freebits=opt.use_kl_freebits,
                  kl_freebits_margin=opt.kl_freebits_margin)
    else:
        kw.update(use_kl_annealing=False)
    loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab, **kw)
    crit = loss.criterion

    class _Crit1(torch.nn.Module):                         # s5
        def forward(self, s, t):
            return crit(s, t).view(1)
    loss.criterion = _Crit1()
    return loss


@contextlib.contextmanager
def inject_eps(eps):
    """Make `Normal.sample()` return loc + scale*eps (detached, like torch.normal: H2)."""
    onmt, _ = import_reference()
    N = onmt.modules.Dists.Normal
    real = N.sample

    def sample(self):
        with torch.no_grad():
            return self.normal.mean + self.normal.scale * eps
    N.sample = sample
    try:
        yield
    finally:
        N.sample = real


@contextlib.contextmanager
def training_shims():
    """s6 + s7: needed only for the sharded (training) loss path."""
    onmt, _ = import_reference()
    real_split = torch.split
    torch.split = lambda t, n, dim=0: tuple(x.clone() for x in real_split(t, n, dim))
    real_cos = onmt.VILoss.compute_cosine
    onmt.VILoss.compute_cosine = lambda p, o: real_cos(p.detach().clone(), o.detach().clone())
    try:
        yield
    finally:
        torch.split = real_split
        onmt.VILoss.compute_cosine = real_cos


@contextlib.contextmanager
def beam_shims():
    """s8: `best_scores_id / num_words` on LongTensors was integer division under torch 0.3 (Beam.py:101)."""
    real = torch.Tensor.__truediv__

    def tdiv(self, other):
        other_float = isinstance(other, float) or (torch.is_tensor(other) and other.is_floating_point())
        if not self.is_floating_point() and not other_float:
            return torch.div(self, other, rounding_mode="floor")
        return real(self, other)
    torch.Tensor.__truediv__ = tdiv
    try:
        yield
    finally:
        torch.Tensor.__truediv__ = real


class Batch(object):
    """The torchtext batch contract of SURVEY.md section 8b."""

    def __init__(self, src, src_len, tgt, tgt_len, indices):
        self.src = (src, src_len)
        self.tgt = (tgt, tgt_len)
        self.indices = indices
        self.batch_size = src.size(1)
