"""mmif -- MI355X-native engine for the image-fusion hot path (host side, Python on PyTorch-ROCm).

torch is used for device memory, streams and torch.distributed only; every FLOP of the hot path is
executed by the hand-written HIP kernels in ../csrc through the C ABI of include/mmif.h."""
from . import _lib  # noqa: F401  (raises if libmmif_hip.so is missing)
from ._lib import IMPL_AUTO, IMPL_MFMA, IMPL_VALU, MmifError, version  # noqa: F401
