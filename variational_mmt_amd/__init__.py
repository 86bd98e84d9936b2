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
    return pkg
