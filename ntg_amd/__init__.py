"""MI355X-native batched trajectory generation behind NTG's C ABI.

The compute path is libntg_amd.so (hand-written HIP for gfx950, C ABI in include/ntg_amd.h).
This package is only host plumbing for tests and bench.py: problem specs, synthetic inputs,
and ctypes bindings that hand device pointers to the library.
"""
from .spec import AV, Spec  # noqa: F401
