"""long_vita_amd — MI355X (gfx950) native hot path of Long-VITA prefill.

csrc/              hand-written HIP kernels + the C ABI (include/vita_hip.h) -> libvita_hip.so
lib.py / ops.py    ctypes binding and tensor-level entry points (no CPU fallback)
the rest           host-side mirror of the reference's operator / plugin interface for this path
                   (same names, argument meaning and error behaviour as long_vita_megatron/*).
"""
__all__ = ["lib", "ops"]
