"""MI355X-native implementation of the variational multimodal NMT (VI_Model1) training step of
iacercalixto/variational_mmt: hand-written HIP kernels (csrc/, C-ABI in include/vmmt.h) driven from Python."""
__version__ = "0.1"


def install_as_onmt(tables=True, legacy_torch_load=True):
    """Make `import onmt` resolve to variational_mmt_amd.onmt: the three lines a maintainer puts at the top of the reference's
    `train_mm_vi_model1.py` / `translate_mm_vi.py` so that everything those drivers import resolves here.

      * `onmt` and its sub-modules (`onmt.io`, `onmt.Models`, `onmt.ModelConstructor`, `onmt.modules` incl. `onmt.modules.SRU.CheckSRU`
        which the reference's own `opts.py:2` needs at import time, `onmt.Utils`, `onmt.translate`, ...), and the class paths pickles
        name (`onmt.Optim.Optim` in checkpoints, `onmt.io.TextDataset.TextDataset` in dataset files);
      * stand-ins for `torchtext.vocab.Vocab` / `torchtext.data.Example` when torchtext 0.2.3 is not installed (pickled in
        checkpoints and dataset files);
      * tables=True: `import tables` (train_mm_vi_model1.py:24, translate_mm_vi.py:18) resolves to the stand-alone HDF5 reader
        `onmt.h5tables` when PyTables is not installed;
      * legacy_torch_load=True: the reference targets torch 0.3.1, where `torch.load(path)` is a plain unpickle; its drivers load
        pickled Python objects that way (datasets, vocabularies, checkpoints holding `opt` and the optimiser object:
        train_mm_vi_model1.py:374,393,544).  torch >= 2.6 refuses such files unless `weights_only=False` is passed, so calls that do
        not say either way get the reference-era default.  Files are as trusted as the reference assumes them to be.
    """
    import importlib
    import sys
    pkg = importlib.import_module("variational_mmt_amd.onmt")
    sys.modules["onmt"] = pkg
    for sub in ("io", "Utils", "Loss", "Models", "ModelConstructor", "Optim", "Trainer", "TrainerMultimodal",
                "VILoss", "modules", "modules.Dists", "modules.SRU", "translate", "translate.Beam", "translate.TranslatorMultimodalVI",
                "h5tables", "bleu", "EarlyStop", "translate.translate_file"):
        sys.modules["onmt." + sub] = importlib.import_module("variational_mmt_amd.onmt." + sub)
    for mod, names in (("Optim", ("Optim", "_ArenaAdam")), ("TrainerMultimodal", ("TrainerMultimodal", "VIStatistics")),
                       ("Trainer", ("Statistics",))):
        m = sys.modules["onmt." + mod]
        for n in names:
            getattr(m, n).__module__ = "onmt." + mod
    # module paths that pickles name: the `.train.N.pt` / `.valid.N.pt` files hold an `onmt.io.TextDataset.TextDataset`
    # (onmt/io/TextDataset.py:16, preprocess.py:97-110) and the driver reads them with a plain torch.load
    # (train_mm_vi_model1.py:372-376).  Synthetic modules, registered in sys.modules only: as attributes of the package the
    # names `TextDataset` / `DatasetBase` must stay what `onmt.io` exports (the class)
    import types
    from .onmt.io import textdata as td
    m_td, m_base = types.ModuleType("onmt.io.TextDataset"), types.ModuleType("onmt.io.DatasetBase")
    m_td.TextDataset = m_base.ONMTDatasetBase = td.TextDataset
    m_base.PAD_WORD, m_base.BOS_WORD, m_base.EOS_WORD, m_base.UNK = td.PAD_WORD, td.BOS_WORD, td.EOS_WORD, 0
    sys.modules["onmt.io.TextDataset"], sys.modules["onmt.io.DatasetBase"] = m_td, m_base
    td.TextDataset.__module__ = "onmt.io.TextDataset"          # datasets written here unpickle in the reference too
    _torchtext_standin()
    if tables:
        import importlib.util
        if "tables" not in sys.modules and importlib.util.find_spec("tables") is None:
            sys.modules["tables"] = sys.modules["onmt.h5tables"]
    if legacy_torch_load:
        _legacy_torch_load()
    return pkg


def _legacy_torch_load():
    import functools

    import torch
    if getattr(torch.load, "__vmmt_legacy__", False):
        return
    real = torch.load

    @functools.wraps(real)
    def load(*a, **kw):
        kw.setdefault("weights_only", False)
        return real(*a, **kw)
    load.__vmmt_legacy__ = True
    torch.load = load


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
