"""Python mirror of the reference's InMemoryStore (src/store/in-memory.js) over libolapgpu.

`HipStore` wraps a device-resident store handle; `Plan` wraps a reusable launch plan that runs on
raw device pointers (torch tensors' data_ptr(), or a HipStore's buffers).  Used by the parity
tests, bench.py and the sharded multi-GPU host.  Method names follow the reference's store.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import DTYPES, DTYPE_NAMES, METHODS, OlapError, check

NP_DTYPES = {"int32": np.int32, "uint32": np.uint32, "float32": np.float32, "float64": np.float64}


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32).ravel())


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())


def _tables(rows, np_dtype, c_type):
    """list of per-dimension arrays -> (keepalive list, C array of pointers)"""
    keep = [np.ascontiguousarray(np.asarray(r, dtype=np_dtype).ravel()) for r in rows]
    keep = [k if k.size else np.zeros(1, dtype=np_dtype) for k in keep]
    arr = (C.POINTER(c_type) * max(len(keep), 1))()
    for i, k in enumerate(keep):
        arr[i] = k.ctypes.data_as(C.POINTER(c_type))
    return keep, arr


def _default_kind(default):
    if isinstance(default, str):
        return capi.DEFAULT_NAN if default == "NaN" else capi.DEFAULT_ZERO
    if default != default:
        return capi.DEFAULT_NAN
    if default == 0:
        return capi.DEFAULT_ZERO
    raise OlapError(capi.ERR_INVALID_DEFAULT, "Invalid default value, only NaN and 0 are supported")


def _method_code(method):
    if method is None:
        return METHODS["sum"]
    if isinstance(method, int):
        return method
    return check_neg(capi.lib().olap_method_from_name(str(method).encode()))


def check_neg(rc):
    if rc < 0:
        raise OlapError(rc, capi.last_error())
    return rc


class Plan:
    """A reusable launch plan (olap_*_plan); run() takes raw device pointers."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    @classmethod
    def drillup(cls, dtype, default, method, old_len, new_len, maps):
        L = capi.lib()
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        h = C.c_void_p()
        check(L.olap_drillup_plan(C.byref(h), DTYPES[dtype], _default_kind(default), _method_code(method), len(ol),
                                  ol.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr))
        return cls(h)

    @classmethod
    def drilldown(cls, dtype, default, method, old_len, new_len, maps, distributions=None):
        L = capi.lib()
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        h = C.c_void_p()
        if distributions is not None:
            d = np.ascontiguousarray(np.asarray(distributions, dtype=np.float64))
            dp, dn = d.ctypes.data_as(capi._pdbl), d.size
        else:
            dp, dn = None, 0
        try:
            m = _method_code(method)
        except OlapError:
            m = METHODS["first"]  # any method other than 'sum' copies (in-memory.js:421-423)
        check(L.olap_drilldown_plan(C.byref(h), DTYPES[dtype], _default_kind(default), m, len(ol),
                                    ol.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr, dp, dn))
        return cls(h)

    @classmethod
    def dice(cls, dtype, default, old_len, new_len, sel):
        L = capi.lib()
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(sel, np.int32, C.c_int32)
        h = C.c_void_p()
        check(L.olap_dice_plan(C.byref(h), DTYPES[dtype], _default_kind(default), len(ol),
                               ol.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr))
        return cls(h)

    @classmethod
    def dice_drillup(cls, dtype, default, method, old_len, mid_len, new_len, sel, maps):
        """Fused dice -> drillUp (one rolled-up dimension)."""
        L = capi.lib()
        ol, ml, nl = _u32(old_len), _u32(mid_len), _u32(new_len)
        keep_s, arr_s = _tables(sel, np.int32, C.c_int32)
        keep_m, arr_m = _tables(maps, np.uint32, C.c_uint32)
        h = C.c_void_p()
        check(L.olap_dice_drillup_plan(C.byref(h), DTYPES[dtype], _default_kind(default), _method_code(method), len(ol),
                                       ol.ctypes.data_as(capi._pu32), ml.ctypes.data_as(capi._pu32),
                                       nl.ctypes.data_as(capi._pu32), arr_s, arr_m))
        return cls(h)

    @classmethod
    def reorder(cls, dtype, default, old_len, perm):
        L = capi.lib()
        ol, p = _u32(old_len), _i32(perm)
        h = C.c_void_p()
        check(L.olap_reorder_plan(C.byref(h), DTYPES[dtype], _default_kind(default), len(ol),
                                  ol.ctypes.data_as(capi._pu32), p.ctypes.data_as(capi._pi32)))
        return cls(h)

    @classmethod
    def load(cls, dtype, my_default, his_default, my_len, his_len, his_to_mine):
        L = capi.lib()
        ml, hl = _u32(my_len), _u32(his_len)
        keep, arr = _tables(his_to_mine, np.int32, C.c_int32)
        h = C.c_void_p()
        check(L.olap_load_plan(C.byref(h), DTYPES[dtype], _default_kind(my_default), _default_kind(his_default),
                               len(ml), ml.ctypes.data_as(capi._pu32), hl.ctypes.data_as(capi._pu32), arr))
        return cls(h)

    @property
    def in_cells(self):
        return int(capi.lib().olap_plan_in_cells(self._h))

    @property
    def out_cells(self):
        return int(capi.lib().olap_plan_out_cells(self._h))

    @property
    def kernel_name(self):
        return capi.lib().olap_plan_kernel_name(self._h).decode()

    def run(self, in_values, in_status, out_values, out_status, stream=None):
        """All four are integer device addresses (0/None = absent); stream = hipStream_t address."""
        check(capi.lib().olap_plan_run(self._h, in_values or None, in_status or None, out_values or None,
                                       out_status or None, stream or None))

    def run_batch(self, in_values, in_status, out_values, out_status, stream=None):
        """The same plan over several buffer pairs in one call (one launch where the plan allows it): lists of device
        addresses; in_status / out_status may be None (no masks) or lists with 0 / None entries."""
        n = len(in_values)

        def ptrs(xs):
            if xs is None:
                return None
            return (C.c_void_p * n)(*[x or None for x in xs])

        check(capi.lib().olap_plan_run_batch(self._h, n, ptrs(in_values), ptrs(in_status), ptrs(out_values), ptrs(out_status),
                                             stream or None))

    def run_batch_rules(self, methods, in_values, in_status, out_values, out_status, stream=None):
        """A drillUp plan over several buffer pairs with a rule each (names or codes): one mixed-rule launch where the
        plan allows it (olap_plan_run_batch_rules)."""
        n = len(in_values)

        def ptrs(xs):
            if xs is None:
                return None
            return (C.c_void_p * n)(*[x or None for x in xs])

        codes = (C.c_int * n)(*[m if isinstance(m, int) else _method_code(m) for m in methods])
        check(capi.lib().olap_plan_run_batch_rules(self._h, n, codes, ptrs(in_values), ptrs(in_status), ptrs(out_values), ptrs(out_status),
                                                   stream or None))

    def status(self):
        check(capi.lib().olap_plan_status(self._h))

    def destroy(self):
        if self._h:
            capi.lib().olap_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipStore:
    """Device-resident store of one measure: the reference's InMemoryStore on an MI355X."""

    def __init__(self, size, type="float32", default=float("nan"), _handle=None):
        self._lib = capi.lib()
        if _handle is not None:
            self._h = _handle
            return
        kind = _default_kind(default)  # in-memory.js:56-57 is checked before :59-60
        if type not in DTYPES:
            raise OlapError(capi.ERR_INVALID_TYPE, "Invalid type")
        h = C.c_void_p()
        check(self._lib.olap_store_create(C.byref(h), int(size), DTYPES[type], kind))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.olap_store_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- properties (in-memory.js:8-28)
    @property
    def size(self):
        return int(self._lib.olap_store_size(self._h))

    @property
    def type(self):
        return DTYPE_NAMES[self._lib.olap_store_dtype(self._h)]

    @property
    def default_is_nan(self):
        return self._lib.olap_store_default(self._h) == capi.DEFAULT_NAN

    @property
    def byte_length(self):
        return int(self._lib.olap_store_byte_length(self._h))

    @property
    def values_ptr(self):
        return self._lib.olap_store_values_ptr(self._h)

    @property
    def status_ptr(self):
        return self._lib.olap_store_status_ptr(self._h)

    @property
    def total(self):
        t = C.c_double()
        check(self._lib.olap_store_total(self._h, C.byref(t)))
        return t.value

    def track_order(self, on=True):
        """olap_store_track_order: keep the reference Map's insertion order (first / last, keys(), serialize())."""
        check(self._lib.olap_store_track_order(self._h, 1 if on else 0))
        return self

    @property
    def order_tracked(self):
        return int(self._lib.olap_store_order_tracked(self._h))

    def count_set(self):
        n = C.c_uint64()
        check(self._lib.olap_store_count_set(self._h, C.byref(n)))
        return n.value

    # ---- data in / out
    def set_data(self, values):
        """`data` setter from a typed numpy array of the store's dtype."""
        v = np.ascontiguousarray(np.asarray(values, dtype=NP_DTYPES[self.type]).ravel())
        check(self._lib.olap_store_set_data(self._h, v.ctypes.data_as(C.c_void_p), v.size))

    def set_data_f64(self, values):
        """`data` setter from JS numbers (float64), converted like a TypedArray store."""
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
        check(self._lib.olap_store_set_data_f64(self._h, v.ctypes.data_as(capi._pdbl), v.size))

    def get_data(self):
        out = np.zeros(max(self.size, 1), dtype=NP_DTYPES[self.type])
        check(self._lib.olap_store_get_data(self._h, out.ctypes.data_as(C.c_void_p)))
        return out[: self.size]

    def get_data_f64(self):
        out = np.zeros(max(self.size, 1), dtype=np.float64)
        check(self._lib.olap_store_get_data_f64(self._h, out.ctypes.data_as(capi._pdbl)))
        return out[: self.size]

    def get_status(self):
        out = np.zeros(max(self.size, 1), dtype=np.int32)
        check(self._lib.olap_store_get_status(self._h, out.ctypes.data_as(capi._pi32)))
        return out[: self.size]

    def keys(self):
        n = C.c_uint64()
        check(self._lib.olap_store_get_keys(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint64)
        check(self._lib.olap_store_get_keys(self._h, out.ctypes.data_as(capi._pu64), n.value, C.byref(n)))
        return out[: n.value]

    def to_sparse(self):
        """(indexes uint32[n], values typed[n]) of the set cells, ascending: the reference's serialised form."""
        n = C.c_uint64()
        check(self._lib.olap_store_to_sparse(self._h, None, None, 0, C.byref(n)))
        idx = np.zeros(max(n.value, 1), dtype=np.uint32)
        vals = np.zeros(max(n.value, 1), dtype=NP_DTYPES[self.type])
        check(self._lib.olap_store_to_sparse(self._h, idx.ctypes.data_as(capi._pu32), vals.ctypes.data_as(C.c_void_p),
                                             n.value, C.byref(n)))
        return idx[: n.value], vals[: n.value]

    @classmethod
    def from_sparse(cls, size, type, default, indexes, values):
        L = capi.lib()
        idx = np.ascontiguousarray(np.asarray(indexes, dtype=np.uint32))
        vals = np.ascontiguousarray(np.asarray(values, dtype=NP_DTYPES[type]))
        h = C.c_void_p()
        check(L.olap_store_from_sparse(C.byref(h), int(size), DTYPES[type], _default_kind(default),
                                       idx.ctypes.data_as(capi._pu32), vals.ctypes.data_as(C.c_void_p), idx.size))
        return cls(0, _handle=h)

    def get_value(self, index):
        v, s = C.c_double(), C.c_int()
        check(self._lib.olap_store_get_value(self._h, int(index), C.byref(v), C.byref(s)))
        return v.value, bool(s.value)

    def set_value(self, index, value):
        if value is None:
            check(self._lib.olap_store_set_value(self._h, int(index), 0.0, 1))
        else:
            check(self._lib.olap_store_set_value(self._h, int(index), float(value), 0))

    def fill(self, value):
        check(self._lib.olap_store_fill(self._h, float(value)))

    def clone(self):
        h = C.c_void_p()
        check(self._lib.olap_store_clone(self._h, C.byref(h)))
        return HipStore(0, _handle=h)

    def totals(self, lens, methods):
        """olap_store_totals: the extended cube of shape (len + 1) per dimension — every marginal of
        getNestedObject(measure, withTotals) — as (float64 values, Int32 mask, launches, bytes read)."""
        ol = _u32(lens)
        m = (C.c_int * max(len(ol), 1))(*[_method_code(x) for x in methods])
        n = int(np.prod([int(l) + 1 for l in ol])) if len(ol) else 1
        vals, stat = np.zeros(n, np.float64), np.zeros(n, np.int32)
        launches, nbytes = C.c_int(), C.c_uint64()
        check(self._lib.olap_store_totals(self._h, len(ol), ol.ctypes.data_as(capi._pu32), m, vals.ctypes.data_as(capi._pdbl),
                                          stat.ctypes.data_as(capi._pi32), C.byref(launches), C.byref(nbytes)))
        return vals, stat, launches.value, nbytes.value

    # ---- bulk operations (names follow in-memory.js)
    def drill_up(self, old_len, new_len, maps, method="sum"):
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        h = C.c_void_p()
        check(self._lib.olap_store_drillup(self._h, C.byref(h), len(ol), ol.ctypes.data_as(capi._pu32),
                                           nl.ctypes.data_as(capi._pu32), arr, _method_code(method)))
        return HipStore(0, _handle=h)

    @staticmethod
    def drill_up_batch(stores, old_len, new_len, maps, method="sum"):
        """drillUp of several measures of a cube by the same maps and rule: one launch when they share cell type,
        default and size (olap_store_drillup_batch); returns the new stores in order."""
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        n = len(stores)
        hs = (C.c_void_p * n)(*[s._h for s in stores])
        outs = (C.c_void_p * n)()
        check(capi.lib().olap_store_drillup_batch(n, hs, outs, len(ol), ol.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr,
                                                  _method_code(method)))
        return [HipStore(0, _handle=C.c_void_p(outs[i])) for i in range(n)]

    @staticmethod
    def drill_up_multi(stores, methods, old_len, new_len, maps):
        """drillUp of the stored measures of a cube, each with its own rule (olap_store_drillup_multi): one mixed-rule
        launch where the roll-up allows it; returns the new stores in order."""
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        n = len(stores)
        hs = (C.c_void_p * n)(*[s._h for s in stores])
        codes = (C.c_int * n)(*[_method_code(m) for m in methods])
        outs = (C.c_void_p * n)()
        check(capi.lib().olap_store_drillup_multi(n, hs, codes, outs, len(ol), ol.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr))
        return [HipStore(0, _handle=C.c_void_p(outs[i])) for i in range(n)]

    def drill_down(self, old_len, new_len, maps, method="sum", distributions=None, integer_measure=False):
        """in-memory.js:336-430.  `integer_measure`: the measure is DECLARED int32 / uint32 but held in float64 cells (what
        the reference's Map does until serialize()): `sum` spreads the integer remainder (:343, :403-417) all the same."""
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        if distributions is not None:
            d = np.ascontiguousarray(np.asarray(distributions, dtype=np.float64))
            dp, dn = d.ctypes.data_as(capi._pdbl), d.size
        else:
            dp, dn = None, 0
        try:
            m = _method_code(method)
        except OlapError:
            m = METHODS["first"]
        if integer_measure:
            m |= capi.DRILLDOWN_INTEGER_MEASURE
        h = C.c_void_p()
        check(self._lib.olap_store_drilldown(self._h, C.byref(h), len(ol), ol.ctypes.data_as(capi._pu32),
                                             nl.ctypes.data_as(capi._pu32), arr, m, dp, dn))
        return HipStore(0, _handle=h)

    def dice(self, old_len, new_len, sel):
        ol, nl = _u32(old_len), _u32(new_len)
        keep, arr = _tables(sel, np.int32, C.c_int32)
        h = C.c_void_p()
        check(self._lib.olap_store_dice(self._h, C.byref(h), len(ol), ol.ctypes.data_as(capi._pu32),
                                        nl.ctypes.data_as(capi._pu32), arr))
        return HipStore(0, _handle=h)

    def dice_drillup(self, old_len, mid_len, new_len, sel, maps, method="sum"):
        ol, ml, nl = _u32(old_len), _u32(mid_len), _u32(new_len)
        keep_s, arr_s = _tables(sel, np.int32, C.c_int32)
        keep_m, arr_m = _tables(maps, np.uint32, C.c_uint32)
        h = C.c_void_p()
        check(self._lib.olap_store_dice_drillup(self._h, C.byref(h), len(ol), ol.ctypes.data_as(capi._pu32),
                                                ml.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr_s, arr_m,
                                                _method_code(method)))
        return HipStore(0, _handle=h)

    def reorder(self, old_len, perm):
        ol, p = _u32(old_len), _i32(perm)
        h = C.c_void_p()
        check(self._lib.olap_store_reorder(self._h, C.byref(h), len(ol), ol.ctypes.data_as(capi._pu32),
                                           p.ctypes.data_as(capi._pi32)))
        return HipStore(0, _handle=h)

    def load(self, other, my_len, his_len, his_to_mine):
        ml, hl = _u32(my_len), _u32(his_len)
        keep, arr = _tables(his_to_mine, np.int32, C.c_int32)
        check(self._lib.olap_store_load(self._h, other._h, len(ml), ml.ctypes.data_as(capi._pu32),
                                        hl.ctypes.data_as(capi._pu32), arr))
