"""olap_store_totals (getNestedObject(measure, withTotals), src/cube.js:421-440): the extended cube must hold, for
every one of the 2^D subsets of dimensions, exactly what the chain drillUp(dim, 'all') over the subset's dimensions
(ascending, each with the measure's rule for that dimension) yields — checked against the oracle run as that chain,
with the typed store's rounding after every step (golden_util.expected_typed)."""
import itertools

import numpy as np
import pytest

from conftest import load_package
from golden_util import expected_typed
from oracle.oracle import OracleStore

pytestmark = pytest.mark.gpu

pkg = load_package()
METHODS = ["sum", "average", "highest", "lowest", "first", "last", "product"]


def chain(vals, type_name, default, lens, methods, subset):
    """oracle: drillUp(dim, 'all') for every dimension of `subset`, ascending; returns (typed values, mask, lens)."""
    cur_lens = list(lens)
    o = OracleStore(len(vals), type_name, default)
    o.set_data(vals)
    ev, es = expected_typed(o)
    for d in sorted(subset):
        new_lens = list(cur_lens)
        new_lens[d] = 1
        maps = [np.zeros(l, np.uint32) if i == d else np.arange(l, dtype=np.uint32) for i, l in enumerate(cur_lens)]
        o = OracleStore(int(np.prod(cur_lens)), type_name, default)
        o.set_data(np.where(es == 2, ev.astype(np.float64), default))
        ev, es = expected_typed(o.drill_up(cur_lens, new_lens, maps, methods[d]))
        cur_lens = new_lens
    return ev, es, cur_lens


def check_totals(lens, type_name, default, methods, seed, frac=0.7, expect_one_launch=None):
    rng = np.random.default_rng(seed)
    n = int(np.prod(lens))
    vals = rng.integers(-6, 7, size=n).astype(np.float64) if type_name != "uint32" else rng.integers(0, 9, size=n).astype(np.float64)
    if type_name.startswith("float"):
        vals = vals / 4.0
    vals = np.where(rng.random(n) < frac, vals, default)
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(vals)
    ext, est, launches, nbytes = g.totals(lens, methods)
    ext_shape = [l + 1 for l in lens]
    ext, est = ext.reshape(ext_shape), est.reshape(ext_shape)
    item = 8 if type_name == "float64" else 4
    primary = default != default and type_name in ("int32", "uint32")
    if expect_one_launch:
        assert launches == 1 and nbytes == n * (item + (4 if primary else 0)), (launches, nbytes)  # the cube is read once
    elif expect_one_launch is False:
        # groups of dimensions fused through LDS (csrc/olap_totals.hip): never more passes than dimensions, each reading
        # the cube-so-far once; only a cube whose INNERMOST dimension exceeds a tile keeps scatter + D stages + export
        if (lens[-1] + 1) * (item + 1) > 150 * 1024:  # not even one workgroup per CU holds a row of the innermost dimension
            assert launches == len(lens) + 2
        else:
            assert 1 <= launches <= len(lens), launches
            ext_cells = int(np.prod([l + 1 for l in lens]))
            assert n * item <= nbytes <= n * (item + (4 if primary else 0)) + (launches - 1) * ext_cells * (item + 1), (launches, nbytes)
    for r in range(len(lens) + 1):
        for subset in itertools.combinations(range(len(lens)), r):
            ev, es, out_lens = chain(vals, type_name, default, lens, methods, subset)
            index = tuple(lens[d] if d in subset else slice(0, lens[d]) for d in range(len(lens)))
            got_v, got_s = ext[index].ravel(), est[index].ravel()
            want = ev.astype(np.float64)
            if default != default:
                want = np.where(es == 2, want, np.nan)
            assert np.array_equal(got_s, es), (subset, methods)
            assert np.array_equal(got_v, want, equal_nan=True), (subset, methods, got_v[:6], want[:6])


@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)])
@pytest.mark.parametrize("method", METHODS)
def test_totals_one_rule_everywhere(type_name, default, method):
    check_totals([4, 3, 5], type_name, default, [method] * 3, seed=hash((type_name, method)) % 1000, expect_one_launch=True)


@pytest.mark.parametrize("seed", range(12))
def test_totals_mixed_rules_lds(seed):
    """A different rule per dimension (average of sums differs from sum of averages on sparse cells: the chain order
    — ascending dimension index — is part of the result)."""
    rng = np.random.default_rng(100 + seed)
    nd = int(rng.integers(1, 6))
    lens = [int(x) for x in rng.integers(1, 6, size=nd)]
    methods = [METHODS[int(i)] for i in rng.integers(0, 7, size=nd)]
    t, d = [("float32", 0.0), ("float32", float("nan")), ("int32", float("nan")), ("float64", float("nan"))][seed % 4]
    check_totals(lens, t, d, methods, seed, frac=[1.0, 0.6, 0.3][seed % 3], expect_one_launch=True)


@pytest.mark.parametrize("lens,methods", [([40, 30, 12], ["sum", "average", "last"]), ([300, 50], ["average", "sum"]), ([7, 6, 5, 4, 3, 2], ["sum"] * 6),
                                          ([3, 5000, 4], ["highest", "sum", "average"]),      # a dimension too long for a tile, between two groups
                                          ([5, 7, 1000], ["first", "product", "lowest"]),     # ragged runs of q in the outer group's tiles
                                          ([6, 5, 4, 3, 7, 2, 3], ["average", "sum", "last", "sum", "highest", "first", "sum"]),
                                          ([2, 3, 13000], ["sum", "sum", "average"]),         # a tile that owns a CU (65 KB)
                                          ([2, 3, 40000], ["sum", "sum", "average"])])         # innermost dimension beyond any tile: round-2 form
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("uint32", float("nan")), ("float64", float("nan"))])
def test_totals_larger_than_lds(lens, methods, type_name, default):
    """Extended cubes above 12288 cells: groups of dimensions fused through LDS, pass after pass."""
    ext = int(np.prod([l + 1 for l in lens]))
    check_totals(lens, type_name, default, methods, seed=len(lens), expect_one_launch=ext <= 12288)


def test_totals_reference_literals():
    """test/cube-accessors.js:41-48: antennas [[1,2],[4,8],[16,32]] -> all x all = 63, rows 3 / 12 / 48, columns 21 / 42."""
    g = pkg.HipStore(6, "uint32", 0.0)
    g.set_data_f64([1, 2, 4, 8, 16, 32])
    ext, est, launches, nbytes = g.totals([3, 2], ["sum", "sum"])
    assert ext.reshape(4, 3).tolist() == [[1, 2, 3], [4, 8, 12], [16, 32, 48], [21, 42, 63]] and launches == 1 and nbytes == 24
    # a zero-dimensional cube: the single cell
    g = pkg.HipStore(1, "float32", 0.0)
    g.set_data_f64([32])
    assert g.totals([], [])[0].tolist() == [32.0]
    with pytest.raises(pkg.OlapError, match="Unsupported aggregation method"):
        g.totals([1], [9])
