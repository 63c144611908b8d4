"""ctypes binding of the CPU oracle (oracle/olap_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.  The product path (olap-in-memory_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libolap_oracle.so")

TYPES = {"int32": 0, "uint32": 1, "float32": 2, "float64": 3}
METHODS = {"sum": 0, "average": 1, "highest": 2, "lowest": 3, "first": 4, "last": 5, "product": 6}
NP_TYPES = {"int32": np.int32, "uint32": np.uint32, "float32": np.float32, "float64": np.float64}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("olap_oracle.c", "olap_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libolap_oracle.so"])
    return _LIB_PATH


def build_flat(force=False):
    """oracle/flat_baseline.c -> libflat_baseline.so (bench.py's dense CPU baselines)."""
    path = os.path.join(_HERE, "libflat_baseline.so")
    src = os.path.join(_HERE, "flat_baseline.c")
    if force or not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libflat_baseline.so"])
    return path


def flat_baseline():
    """ctypes view of flat_baseline.c: (lib, max threads)."""
    L = C.CDLL(build_flat())
    L.flat_drillup_sum.restype = C.c_double
    L.flat_drillup_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    L.flat_max_threads.restype = C.c_int
    return L, int(L.flat_max_threads())


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u64, i32, dbl = C.c_void_p, C.c_uint64, C.c_int, C.c_double
        pu32, pi32, pdbl, pu64, pu8 = (C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint8))
        sig = {
            "oracle_store_new": (vp, [u64, i32, i32]),
            "oracle_store_free": (None, [vp]),
            "oracle_store_clone": (vp, [vp]),
            "oracle_size": (u64, [vp]),
            "oracle_type": (i32, [vp]),
            "oracle_default_is_nan": (i32, [vp]),
            "oracle_num_keys": (u64, [vp]),
            "oracle_entries": (None, [vp, pu64, pdbl]),
            "oracle_dense": (None, [vp, pdbl, pu8]),
            "oracle_total": (dbl, [vp]),
            "oracle_get_value": (dbl, [vp, u64]),
            "oracle_set_value": (None, [vp, u64, dbl]),
            "oracle_unset_value": (None, [vp, u64]),
            "oracle_set_data": (None, [vp, pdbl]),
            "oracle_fill": (None, [vp, dbl]),
            "oracle_fill_seeded": (None, [vp, C.c_uint32, dbl]),
            "oracle_drillup": (vp, [vp, i32, pu32, pu32, pu32, i32]),
            "oracle_drilldown": (vp, [vp, i32, pu32, pu32, pu32, i32, pdbl, u64, C.POINTER(C.c_int64)]),
            "oracle_dice": (vp, [vp, i32, pu32, pu32, pi32]),
            "oracle_reorder": (vp, [vp, i32, pu32, pi32]),
            "oracle_load": (None, [vp, vp, i32, pu32, pu32, pi32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32).ravel())


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _cat(tables, dtype):
    if len(tables) == 0:
        return np.zeros(1, dtype=dtype)
    flat = [np.asarray(t, dtype=dtype).ravel() for t in tables]
    out = np.concatenate(flat) if flat else np.zeros(0, dtype=dtype)
    return np.ascontiguousarray(out if out.size else np.zeros(1, dtype=dtype))


class OracleStore:
    """Mirror of the reference's InMemoryStore (src/store/in-memory.js) on the C oracle."""

    def __init__(self, size, type="float32", default=float("nan"), _handle=None):
        self._lib = lib()
        if _handle is not None:
            self._h = _handle
        else:
            if type not in TYPES:
                raise ValueError("Invalid type")
            is_nan = default != default
            if not is_nan and default != 0:
                raise ValueError("Invalid default value, only NaN and 0 are supported")
            self._h = self._lib.oracle_store_new(int(size), TYPES[type], int(is_nan))

    def __del__(self):
        try:
            if self._h:
                self._lib.oracle_store_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def size(self):
        return int(self._lib.oracle_size(self._h))

    @property
    def type(self):
        return [k for k, v in TYPES.items() if v == self._lib.oracle_type(self._h)][0]

    @property
    def default_is_nan(self):
        return bool(self._lib.oracle_default_is_nan(self._h))

    @property
    def num_keys(self):
        return int(self._lib.oracle_num_keys(self._h))

    def entries(self):
        n = self.num_keys
        keys = np.zeros(max(n, 1), dtype=np.uint64)
        vals = np.zeros(max(n, 1), dtype=np.float64)
        self._lib.oracle_entries(self._h, _ptr(keys, C.c_uint64), _ptr(vals, C.c_double))
        return keys[:n], vals[:n]

    def dense(self):
        n = self.size
        vals = np.zeros(max(n, 1), dtype=np.float64)
        pres = np.zeros(max(n, 1), dtype=np.uint8)
        self._lib.oracle_dense(self._h, _ptr(vals, C.c_double), _ptr(pres, C.c_uint8))
        return vals[:n], pres[:n].astype(bool)

    def total(self):
        return float(self._lib.oracle_total(self._h))

    def get(self, i):
        return float(self._lib.oracle_get_value(self._h, int(i)))

    def set(self, i, v):
        if v is None:
            self._lib.oracle_unset_value(self._h, int(i))
        else:
            self._lib.oracle_set_value(self._h, int(i), float(v))

    def set_data(self, values):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
        if v.size != self.size:
            raise ValueError(f"value length is invalid: {self.size} !== {v.size}")
        self._lib.oracle_set_data(self._h, _ptr(v, C.c_double))

    def fill(self, v):
        self._lib.oracle_fill(self._h, float(v))

    def fill_seeded(self, seed, frac):
        self._lib.oracle_fill_seeded(self._h, int(seed) & 0xFFFFFFFF, float(frac))

    def clone(self):
        return OracleStore(0, _handle=self._lib.oracle_store_clone(self._h))

    def drill_up(self, old_len, new_len, maps, method="sum"):
        if method not in METHODS:
            raise ValueError(f"Unsupported aggregation method: {method}")
        ol, nl, mp = _u32(old_len), _u32(new_len), _cat(maps, np.uint32)
        h = self._lib.oracle_drillup(self._h, len(old_len), _ptr(ol, C.c_uint32), _ptr(nl, C.c_uint32),
                                     _ptr(mp, C.c_uint32), METHODS[method])
        return OracleStore(0, _handle=h)

    def drill_down(self, old_len, new_len, maps, method="sum", distributions=None):
        ol, nl, mp = _u32(old_len), _u32(new_len), _cat(maps, np.uint32)
        missing = C.c_int64(-1)
        if distributions is not None:
            d = np.ascontiguousarray(np.asarray(distributions, dtype=np.float64))
            dp, dn = _ptr(d, C.c_double), d.size
        else:
            dp, dn = None, 0
        h = self._lib.oracle_drilldown(self._h, len(old_len), _ptr(ol, C.c_uint32), _ptr(nl, C.c_uint32),
                                       _ptr(mp, C.c_uint32), METHODS.get(method, 4), dp, dn, C.byref(missing))
        if not h:
            raise ValueError(f"distribution missing for index {missing.value}")
        return OracleStore(0, _handle=h)

    def dice(self, old_len, new_len, sel):
        ol, nl, s = _u32(old_len), _u32(new_len), _cat(sel, np.int32)
        h = self._lib.oracle_dice(self._h, len(old_len), _ptr(ol, C.c_uint32), _ptr(nl, C.c_uint32), _ptr(s, C.c_int32))
        return OracleStore(0, _handle=h)

    def reorder(self, old_len, perm):
        ol, p = _u32(old_len), _i32(perm)
        h = self._lib.oracle_reorder(self._h, len(old_len), _ptr(ol, C.c_uint32), _ptr(p, C.c_int32))
        return OracleStore(0, _handle=h)

    def load(self, other, my_len, his_len, his_to_mine):
        ml, hl, m = _u32(my_len), _u32(his_len), _cat(his_to_mine, np.int32)
        self._lib.oracle_load(self._h, other._h, len(my_len), _ptr(ml, C.c_uint32), _ptr(hl, C.c_uint32), _ptr(m, C.c_int32))

    def typed(self):
        """Dense values converted the way the reference's serialize() would coerce them
        (TypedArray conversion, in-memory.js:77-92) plus the Int32 status mask (0x2 = set)."""
        vals, pres = self.dense()
        return to_typed(vals, self.type), np.where(pres, 2, 0).astype(np.int32)


def to_typed(vals, type_name):
    """ECMAScript TypedArray element conversion of float64 numbers."""
    vals = np.asarray(vals, dtype=np.float64)
    if type_name == "float64":
        return vals.copy()
    if type_name == "float32":
        with np.errstate(over="ignore"):
            return vals.astype(np.float32)
    # ToInt32 / ToUint32: NaN, +-Inf -> 0; truncate; modulo 2^32
    finite = np.isfinite(vals)
    t = np.where(finite, np.trunc(vals), 0.0)
    m = np.mod(t, 4294967296.0)
    u = m.astype(np.uint64).astype(np.uint32)
    return u.view(np.int32).copy() if type_name == "int32" else u
