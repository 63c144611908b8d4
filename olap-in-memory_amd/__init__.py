"""olap-in-memory on MI355X: Python-side binding of libolapgpu (tests, bench, multi-GPU host).

The product's host language is Node.js (olap-in-memory_amd/js, bound through the N-API addon in
olap-in-memory_amd/napi); this package is the ctypes view of the same C ABI (include/olap_hip.h)
used by pytest, bench.py and the torch.distributed sharding host.  Importing it never builds or
falls back to anything: if lib/libolapgpu.so is missing, binding raises.
"""
from . import capi  # noqa: F401
from .capi import OlapError, lib, lib_path  # noqa: F401
from .hipstore import HipStore, Plan  # noqa: F401

__all__ = ["capi", "OlapError", "lib", "lib_path", "HipStore", "Plan"]
