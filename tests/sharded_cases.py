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
    # the round-2 failure: cancellation ACROSS ranks.  The reference adds every contribution in
    # float64 (in-memory.js:282-290), so 2^24 + 1 - 2^24 = 1 and the cell is set; partials rounded to Float32 per rank
    # gave 0 and unset (2^24 + 1 rounds to 2^24).  Rows split 2 + 2 (two ranks) and 2 + 1 + 1 (three).
    "f32_cancel": dict(lens=[4, 1], dtype="float32", default=0.0, row_map=[0, 0, 0, 0], groups=1,
                       literal=[16777216.0, 1.0, -16777216.0, 0.0]),
    "f32_cancel_nan": dict(lens=[4, 3], dtype="float32", default=float("nan"), row_map=[0, 0, 0, 0], groups=1,
                           literal=[16777216.0, np.nan, 3.0, 1.0, np.nan, 2.0 ** -30, -16777216.0, np.nan, -3.0, np.nan, np.nan, np.nan]),
    # integer sums pass 2^32 inside the float64 accumulator: `sum` wraps modulo 2^32 at storage on every path, `average`
    # divides the exact sum first (4e9 + 4e9 over two cells is 4e9, not (8e9 mod 2^32) / 2)
    "u32_average_overflow": dict(lens=[4, 2], dtype="uint32", default=0.0, row_map=[0, 0, 0, 0], groups=1,
                                 literal=[4e9, 1.0, 4e9, 0.0, 4e9, 3.0, 0.0, 7.0], skip=("product",)),
    # a wide cube: the row regime with 16-byte lanes writes its float64 partials as two 16-byte stores per lane
    "f32_zero_wide": dict(lens=[6, 1024], dtype="float32", default=0.0, row_map=[0, 1, 0, 1, 1, 0], groups=2, frac=0.9, seed=19,
                          only=("sum", "average")),
    "f32_nan_ragged": dict(lens=[5, 1031], dtype="float32", default=float("nan"), row_map=[0, 0, 1, 0, 1], groups=2, frac=0.7, seed=20,
                           only=("sum", "average")),
    # few outputs, long groups: the cooperative reduce regime emits the partials from its merge kernels
    "f32_zero_tall": dict(lens=[3000, 3], dtype="float32", default=0.0, row_map=[0] * 3000, groups=1, frac=0.9, seed=21,
                          only=("sum", "average")),
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
        # Float32-representable cells of mixed sign and magnitude, m * 2^-e with |m| < 2^24 and e in 0..20: their sums are
        # NOT exact in Float32 (the partials of a rank round differently from the whole), but every sum of fewer than 2^8
        # of them is a multiple of 2^-20 below 2^32 — exact in float64 in ANY order.  So a sharded sum / average that
        # keeps float64 partials and rounds once agrees with the one-device result and the oracle BIT FOR BIT.
        m = rng.integers(-(2 ** 24) + 1, 2 ** 24, size=n).astype(np.float64)
        e = rng.integers(0, 21, size=n)
        if n > 10000:  # (tall cubes add thousands of cells per output: keep |sum| * 2^20 < 2^53)
            m = rng.integers(-(2 ** 16) + 1, 2 ** 16, size=n).astype(np.float64)
        v = m * np.exp2(-e.astype(np.float64))
        assert np.array_equal(v.astype(np.float32).astype(np.float64), v)
    keep = rng.random(n) < case["frac"]
    return np.where(keep, v, np.nan if default_nan else 0.0)


def methods_of(case):
    return [m for m in METHODS if m not in case.get("skip", ()) and m in case.get("only", METHODS)]
