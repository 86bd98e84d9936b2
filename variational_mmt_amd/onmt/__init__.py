"""Host-side mirror of the reference's `onmt` surface for the VI_Model1 training step.
`variational_mmt_amd.install_as_onmt()` registers it as the top-level `onmt` package so that a driver written against
the reference (`import onmt; onmt.ModelConstructor.make_vi_model_mmt(...)`) and its checkpoints' pickled
`onmt.Optim.Optim` resolve here."""
from . import io, Utils, bleu, EarlyStop as _ES, h5tables, Loss, Models, translate, ModelConstructor, Optim as _OptimMod, Trainer, TrainerMultimodal as _TM, VILoss  # noqa
from .Optim import Optim  # noqa: F401
from .Trainer import Statistics  # noqa: F401
from .TrainerMultimodal import TrainerMultimodal, VIStatistics  # noqa: F401
