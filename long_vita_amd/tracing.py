"""roctx ranges per phase / per layer, behind VITA_DEBUG (SURVEY.md §5 "tracing": the reference marks its phases with Megatron's timers,
M/training/training.py `timers(...)`; on this path the marks are roctx ranges that `rocprofv3 --marker-trace --kernel-trace` lines up with
the kernels).

Off (the default) every call is one attribute test; on, `librocprofiler-sdk-roctx.so` (else `libroctx64.so`) is loaded from the ROCm the process already uses — it is a
tracing dependency only, nothing on the compute path touches it, and a missing library turns the ranges into no-ops with one warning.

    VITA_DEBUG=1 rocprofv3 --marker-trace --kernel-trace --stats -d out -- python bench.py --steps 1 --warmup 0
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import warnings

ENABLED = os.environ.get("VITA_DEBUG", "0") not in ("", "0")
_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    # rocprofv3 records the ranges of rocprofiler-sdk's roctx library; the roctracer-era libroctx64 is the fallback (older tools)
    for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "libroctx64.so.4",
                 "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.argtypes = []
            lib.roctxRangePop.restype = ctypes.c_int
            lib.roctxMarkA.argtypes = [ctypes.c_char_p]
            lib.roctxMarkA.restype = None
            _lib = lib
            return _lib
        except (OSError, AttributeError):
            continue
    warnings.warn("VITA_DEBUG is set but libroctx64.so could not be loaded: ranges are no-ops")
    return None


def push(name: str) -> None:
    if ENABLED:
        lib = _load()
        if lib is not None:
            lib.roctxRangePushA(name.encode())


def pop() -> None:
    if ENABLED:
        lib = _load()
        if lib is not None:
            lib.roctxRangePop()


def mark(name: str) -> None:
    if ENABLED:
        lib = _load()
        if lib is not None:
            lib.roctxMarkA(name.encode())


@contextlib.contextmanager
def _range(name: str):
    push(name)
    try:
        yield
    finally:
        pop()


_NULL = contextlib.nullcontext()


def range(name: str):                                                 # noqa: A001  (the roctx word for it)
    """`with tracing.range("layer 3"):` — a null context when VITA_DEBUG is off."""
    return _range(name) if ENABLED else _NULL


def instrument_functions(namespace: dict) -> None:
    """Wrap forward / backward of every torch.autograd.Function in `namespace` in a roctx range named Class.forward / Class.backward
    (the module path: Megatron's autograd re-enters these Functions, also inside tensor_parallel.checkpoint's recompute).  Called at
    import of autograd_fns; does nothing unless VITA_DEBUG is set, so the default path carries no wrapper at all."""
    if not ENABLED:
        return
    import functools
    import torch
    for name, cls in list(namespace.items()):
        if not (isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function):
            continue
        for meth in ("forward", "backward"):
            fn = cls.__dict__.get(meth)
            if fn is None:
                continue
            raw = fn.__func__ if isinstance(fn, staticmethod) else fn
            label = f"{name}.{meth}"

            def make(raw=raw, label=label):
                @functools.wraps(raw)
                def wrapped(*a, **k):
                    push(label)
                    try:
                        return raw(*a, **k)
                    finally:
                        pop()
                return wrapped
            setattr(cls, meth, staticmethod(make()))
