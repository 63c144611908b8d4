"""Cases shared by tests/_sharded_worker.py (every rank) and tests/test_sharded_gloo.py (the checker):
one sharded drillUp of dimension 0 per store kind the reference's constructor allows
(in-memory.js:51-60: NaN is the DEFAULT default; int32 / uint32 / float32 / float64 cells)."""
import numpy as np

METHODS = ("sum", "average", "highest", "lowest", "first", "last", "product")

_ROWS7 = [0, 1, 0, 2, 1, 0, 2]

CASES = {
    # name: lens, cell type, default, row -> group map of dimension 0, groups, set fraction, value kind
    "f32_zero": dict(lens=[7, 6, 10], dtype="float32", default=0.0, row_map=_ROWS7, groups=3, frac=0.4, seed=11),
    "f32_zero_full": dict(lens=[7, 6, 10], dtype="float32", default=0.0, row_map=_ROWS7, groups=3, frac=1.0, seed=12),
    "f32_nan": dict(lens=[7, 6, 10], dtype="float32", default=float("nan"), row_map=_ROWS7, groups=3, frac=0.4, seed=13),
    "f64_nan": dict(lens=[7, 6, 10], dtype="float64", default=float("nan"), row_map=_ROWS7, groups=3, frac=0.6, seed=14),
    "i32_nan": dict(lens=[7, 6, 10], dtype="int32", default=float("nan"), row_map=_ROWS7, groups=3, frac=0.5, seed=15),
    "i32_zero": dict(lens=[7, 6, 10], dtype="int32", default=0.0, row_map=_ROWS7, groups=3, frac=0.5, seed=16),
    "u32_zero": dict(lens=[7, 6, 10], dtype="uint32", default=0.0, row_map=_ROWS7, groups=3, frac=0.7, seed=17, big=True,
                     skip=("product",)),  # products of 1e9-sized cells exceed 2^53: not integers in the reference either
    # the round-1 failure: a NaN default and output cells only ONE rank contributes to
    # (two ranks own rows 0-1 and 2-3): sum must be [1, 7, 5], never NaN
    "f32_nan_disjoint": dict(lens=[4, 3], dtype="float32", default=float("nan"), row_map=[0, 0, 0, 0], groups=1,
                             literal=[1, np.nan, 2, np.nan, np.nan, np.nan, np.nan, 3, np.nan, np.nan, 4, 3]),
    # more ranks than rows on some worlds; one group per row pair
    "f32_zero_short": dict(lens=[2, 5], dtype="float32", default=0.0, row_map=[0, 0], groups=1, frac=0.8, seed=18),
}


def case_data(case):
    """float64 `data` of the whole cube (unset cells hold the default), identical on every rank."""
    if "literal" in case:
        return np.asarray(case["literal"], np.float64)
    n = int(np.prod(case["lens"]))
    rng = np.random.default_rng(case["seed"])
    default_nan = case["default"] != case["default"]
    integer = case["dtype"] in ("int32", "uint32")
    if case.get("big"):
        v = rng.integers(1, 1_000_000_000, size=n).astype(np.float64)  # three of them still fit 32 bits
    elif integer:
        lo = 0 if case["dtype"] == "uint32" else -4
        v = rng.integers(lo, 6, size=n).astype(np.float64)
    else:
        v = rng.integers(-8, 9, size=n).astype(np.float64) / 4.0  # sums are exact in float32 whatever the order
    keep = rng.random(n) < case["frac"]
    return np.where(keep, v, np.nan if default_nan else 0.0)


def methods_of(case):
    return [m for m in METHODS if m not in case.get("skip", ())]
