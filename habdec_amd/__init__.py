"""habdec_amd -- MI355X-native RTTY demodulation behind habdec's `Decoder<T>` push-samples API.

The product is the C-ABI shared library `libhabdec_amd.so` (HIP kernels for gfx950 + host engine, see
include/habdec_amd.h) and the source-compatible C++ facade in habdec_amd/include/habdec/.  This Python package
is only plumbing for tests and benchmarks: a ctypes binding (`habdec_amd.capi`), a thin `Engine` wrapper, and the
synthetic IQ generator.  There is no Python or CPU implementation of the data path: if the library is missing,
`lib()` raises.
"""
from .capi import lib, LIB_PATH, HabdecError  # noqa: F401
from .engine import Engine, EngineConfig, IqFiles  # noqa: F401

__all__ = ["lib", "LIB_PATH", "HabdecError", "Engine", "EngineConfig", "IqFiles"]
