"""vitlens_hip — MI355X-native kernels for the ViT-Lens contrastive hot path.

Thin Python face of libvitlens_hip.so (C ABI in include/vitlens_hip.h).  PyTorch is used
only for device memory, streams and torch.distributed; every FLOP on the hot path runs in
the hand-written HIP kernels under csrc/.  There is NO CPU or eager fallback: importing
`vitlens_hip.ops` without the built library, or calling an op on a non-GPU tensor, raises.
"""
from ._lib import lib_path, load_library, LibraryNotBuilt  # noqa: F401

__all__ = ["lib_path", "load_library", "LibraryNotBuilt"]
