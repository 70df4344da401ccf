"""MI355X-native (gfx950) volume-rendering hot path of Robust e-NeRF.

Hand-written HIP kernels behind a C ABI (include/ren_amd.h, csrc/libren_amd.so); PyTorch-ROCm
only owns device memory, streams and torch.distributed.  No CPU fallback: importing the ops on a
machine without the built library raises.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "engine"]
