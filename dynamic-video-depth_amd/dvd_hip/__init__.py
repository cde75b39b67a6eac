"""dvd_hip -- MI355X-native (gfx950) implementation of the dynamic-video-depth
test-time-optimisation inner loop.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).
Compute: hand-written HIP kernels in csrc/, reached through the C ABI declared
in include/dvd_hip.h and bound with ctypes in `dvd_hip._lib`.

Importing the package does not load the library; the first operator call does,
and raises RuntimeError if libdvd_hip.so is missing (there is no CPU fallback).
"""

__all__ = ['synthetic']
__version__ = '0.1.0'
