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
from .layout import Buf, Dims, KPAD, PAD, SEG_ALIGN, _ru  # noqa: F401
from .core import Engine  # noqa: F401
from .workspace import Workspace  # noqa: F401
