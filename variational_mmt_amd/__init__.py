"""MI355X-native implementation of the variational multimodal NMT (VI_Model1) training step of
iacercalixto/variational_mmt: hand-written HIP kernels (csrc/, C-ABI in include/vmmt.h) driven from Python."""
__version__ = "0.1"
