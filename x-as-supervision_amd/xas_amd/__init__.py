"""xas_amd - host side of the MI355X-native X-as-Supervision training step.

Python binds libxas_hip.so (the C ABI in include/xas_hip.h) with ctypes; PyTorch is used
for device memory, streams, autograd bookkeeping and torch.distributed only.  There is no
CPU fallback: every op raises if the library is missing or a tensor is not on the GPU.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
