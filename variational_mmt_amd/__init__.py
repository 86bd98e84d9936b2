"""MI355X-native implementation of the variational multimodal NMT (VI_Model1) training step of
iacercalixto/variational_mmt: hand-written HIP kernels (csrc/, C-ABI in include/vmmt.h) driven from Python."""
__version__ = "0.1"


def install_as_onmt():
    """Make `import onmt` resolve to variational_mmt_amd.onmt (drop-in for drivers and for checkpoint pickles that
    name `onmt.Optim.Optim`)."""
    import importlib
    import sys
    pkg = importlib.import_module("variational_mmt_amd.onmt")
    sys.modules["onmt"] = pkg
    for sub in ("io", "Utils", "Loss", "Models", "ModelConstructor", "Optim", "Trainer", "TrainerMultimodal", "VILoss",
                "modules", "modules.Dists", "translate", "translate.Beam", "translate.TranslatorMultimodalVI", "h5tables", "bleu", "EarlyStop", "translate.translate_file"):
        sys.modules["onmt." + sub] = importlib.import_module("variational_mmt_amd.onmt." + sub)
    for mod, names in (("Optim", ("Optim", "_ArenaAdam")), ("TrainerMultimodal", ("TrainerMultimodal", "VIStatistics")),
                       ("Trainer", ("Statistics",))):
        m = sys.modules["onmt." + mod]
        for n in names:
            getattr(m, n).__module__ = "onmt." + mod
    _torchtext_standin()
    return pkg


def _torchtext_standin():
    """checkpoints pickle `torchtext.vocab.Vocab` objects (checkpoint['vocab'], TrainerMultimodal.py:583; IO.py:64-75) and the
    dataset files `torchtext.data.Example`s.  When torchtext (0.2.3, requirements.txt) is not installed, register the
    classes of variational_mmt_amd.onmt.io under those names so that a plain `torch.load(opt.train_from)` of the driver works."""
    import importlib.util
    import sys
    import types
    if "torchtext" in sys.modules or importlib.util.find_spec("torchtext") is not None:
        return
    from .onmt.io import textdata as td
    tt = types.ModuleType("torchtext")
    tt.__path__ = []
    vocab, data, example = types.ModuleType("torchtext.vocab"), types.ModuleType("torchtext.data"), types.ModuleType("torchtext.data.example")
    data.__path__ = []
    vocab.Vocab = td.Vocab
    data.Example = example.Example = td.Example
    data.Field, data.Dataset, data.Batch = td.Field, td.TextDataset, td.Batch
    tt.vocab, tt.data = vocab, data
    data.example = example
    tt.__vmmt_standin__ = True
    sys.modules.update({"torchtext": tt, "torchtext.vocab": vocab, "torchtext.data": data, "torchtext.data.example": example})
