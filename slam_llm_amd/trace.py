"""roctx ranges around the phases of the training step (mel / encoder / projector / llm_fwd / llm_bwd / projector_bwd / encoder_bwd /
optimizer / allreduce), so that `rocprofv3 --marker-trace --kernel-trace` cuts a step by phase (tracing row: the
reference has torch.profiler + FlopMeasure hooks, utils/train_utils.py:81-95; this build's counterpart is rocprofv3 over roctx).

Off by default: `SLAM_ROCTX=1` loads librocprofiler-sdk-roctx (ROCm 7; libroctx64 as the fallback name) through ctypes and pushes /
pops ranges; without it `phase()` is a no-op context manager that costs one attribute test.  Host-side ranges only: they mark where
the launches of a phase are ISSUED; rocprofv3 correlates the kernels launched inside a range with it."""
from __future__ import annotations

import contextlib
import ctypes
import os

_lib = None
ENABLED = os.environ.get("SLAM_ROCTX", "0") == "1"
if ENABLED:
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _lib = ctypes.CDLL(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", name))
            break
        except OSError:
            continue
    if _lib is None:
        raise ImportError("SLAM_ROCTX=1 but neither librocprofiler-sdk-roctx.so nor libroctx64.so could be loaded from $ROCM_PATH/lib")
    _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
    _lib.roctxRangePushA.restype = ctypes.c_int
    _lib.roctxRangePop.restype = ctypes.c_int
    _lib.roctxMarkA.argtypes = [ctypes.c_char_p]

_NULL = contextlib.nullcontext()


class _Range:
    __slots__ = ("name",)

    def __init__(self, name: str):
        self.name = name.encode()

    def __enter__(self):
        _lib.roctxRangePushA(self.name)

    def __exit__(self, *exc):
        _lib.roctxRangePop()
        return False


def phase(name: str):
    """`with trace.phase("llm_fwd"): ...`"""
    return _Range("slam/" + name) if ENABLED else _NULL


def push(name: str):
    if ENABLED:
        _lib.roctxRangePushA(("slam/" + name).encode())


def pop():
    if ENABLED:
        _lib.roctxRangePop()


def mark(name: str):
    if ENABLED:
        _lib.roctxMarkA(("slam/" + name).encode())
