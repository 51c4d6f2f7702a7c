"""wast3d_amd — MI355X-native (gfx950) implementation of the WaSt3D style-loop hot path.

Host side is PyTorch-ROCm (device memory, streams, torch.distributed); every kernel is hand-written HIP
behind the C ABI of include/wast3d_hip.h (libwast3d_hip.so, loaded with ctypes).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
