"""MI355X-native FLE / dictionary decode + predicate -> selection bitmap (libips_hip.so).

The directory name carries a hyphen (it mirrors the reference repository's name), so the package
is loaded by path: ``from __graft_entry__ import load_package; ips = load_package()``.

    ips.capi   ctypes binding of the C-ABI in include/ips.h (torch tensors in, torch tensors out)
    ips.synth  deterministic synthetic columns of SURVEY.md section 8(d)
    ips.sharding  row-stripe sharding + all-gather of bitmap words (RCCL / gloo)
    ips.q6     the TPC-H-Q6-shaped three-column conjunction of BASELINE configs[4]
    csrc/      hand-written HIP kernels + the extern "C" boundary
    host/      C++ facade with the reference's class names on top of the C-ABI

There is no CPU fallback: every entry point raises if libips_hip.so is missing or fails.
"""
from . import capi, q6, sharding, synth  # noqa: F401

__all__ = ["capi", "q6", "sharding", "synth"]
