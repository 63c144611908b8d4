"""Helpers shared by the parity tests: golden-vector decoding (tests/golden/*.json)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_SPECIAL = {"NaN": float("nan"), "Infinity": float("inf"), "-Infinity": float("-inf"), "-0": -0.0}


def dec_num(v):
    if isinstance(v, str):
        return _SPECIAL[v]
    return float(v)


def dec_store(d):
    """-> (size, keys uint64[n] in insertion order, values float64[n])"""
    if "iota" in d:
        n = d["iota"]
        return d["size"], np.arange(n, dtype=np.uint64), np.full(n, dec_num(d["value"]))
    keys = np.asarray(d["keys"], dtype=np.uint64)
    vals = np.asarray([dec_num(v) for v in d["values"]], dtype=np.float64)
    return d["size"], keys, vals


def load_cases(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)["cases"]


def default_of(case, key="default"):
    return float("nan") if case[key] == "NaN" else 0.0


def same_f64(a, b):
    """Bitwise float64 equality, except that any NaN equals any NaN."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(nan | (a.view(np.uint64) == b.view(np.uint64))))


def mulberry32_stream(seed, n):
    """First n draws of the mulberry32 stream used by oracle/gen_golden.js (vectorised)."""
    i = np.arange(1, n + 1, dtype=np.uint64)
    a = ((np.uint64(seed) + i * np.uint64(0x6D2B79F5)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    with np.errstate(over="ignore"):
        t = (a ^ (a >> np.uint32(15))) * (np.uint32(1) | a)
        t = (t + ((t ^ (t >> np.uint32(7))) * (np.uint32(61) | t))) ^ t
        r = t ^ (t >> np.uint32(14))
    return r.astype(np.float64) / 4294967296.0


def mulberry32_at(seed, positions):
    """mulberry32 draws at the given 1-based stream positions (uint64 array), as olap_fill_seeded addresses them."""
    i = np.asarray(positions, dtype=np.uint64)
    a = ((np.uint64(seed) + i * np.uint64(0x6D2B79F5)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    with np.errstate(over="ignore"):
        t = (a ^ (a >> np.uint32(15))) * (np.uint32(1) | a)
        t = (t + ((t ^ (t >> np.uint32(7))) * (np.uint32(61) | t))) ^ t
        r = t ^ (t >> np.uint32(14))
    return r.astype(np.float64) / 4294967296.0


def config_cube(n_cells, seed=20240807, frac=1.0):
    """SURVEY §8(d) synthetic cube: values fround(0.5+u1), cell kept iff u2 < frac.
    Returns (float32 values with 0 where unset, bool present)."""
    u = mulberry32_stream(seed, 2 * n_cells)
    vals = (0.5 + u[0::2]).astype(np.float32)
    keep = u[1::2] < frac
    return np.where(keep, vals, np.float32(0)), keep


# ---- what a TYPED store holds for an oracle (float64 Map) result -------------------------------
# The reference keeps float64 numbers until serialize() (in-memory.js:77-92); this implementation
# stores cells in the declared type after every operation, so an expectation is
# the oracle's result after TypedArray conversion, and a result that converts to the default is unset.
# tests/test_typed_storage_differences.py pins where that differs from the reference's getData().
def is_default_typed(vals, type_name, default_is_nan):
    if type_name in ("float32", "float64"):
        return np.isnan(vals) if default_is_nan else (vals == 0)
    return np.zeros(vals.shape, dtype=bool) if default_is_nan else (vals == 0)


def expected_typed(ostore):
    """Oracle store -> (typed values with the default in unset cells, Int32 status mask)."""
    from oracle.oracle import to_typed

    vals, pres = ostore.dense()
    t = to_typed(vals, ostore.type)
    nan = ostore.default_is_nan
    pres = pres & ~is_default_typed(t, ostore.type, nan)
    if ostore.type in ("float32", "float64"):
        dflt = np.nan if nan else 0.0
    else:
        dflt = 0
    t = np.where(pres, t, np.asarray(dflt, dtype=t.dtype))
    return t, np.where(pres, 2, 0).astype(np.int32)


def same_typed(a, b):
    if a.dtype.kind == "f":
        u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        nan = np.isnan(a) & np.isnan(b)
        return bool(np.all(nan | (a.view(u) == b.view(u))))
    return bool(np.array_equal(a, b))
