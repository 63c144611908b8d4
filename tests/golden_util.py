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


def config_cube(n_cells, seed=20240807, frac=1.0):
    """SURVEY §8(d) synthetic cube: values fround(0.5+u1), cell kept iff u2 < frac.
    Returns (float32 values with 0 where unset, bool present)."""
    u = mulberry32_stream(seed, 2 * n_cells)
    vals = (0.5 + u[0::2]).astype(np.float32)
    keep = u[1::2] < frac
    return np.where(keep, vals, np.float32(0)), keep
