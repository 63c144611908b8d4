"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bit-exact for status masks and integer cells; float cells are compared bit-exactly too, because
the kernels accumulate in float64 in the reference's order (ascending flat index) and round once.
Where a golden input was inserted in a shuffled order (the reference's first/last follow Map
insertion order, which a dense buffer cannot represent), the expectation is the oracle run on the
same cells inserted ascending; this is stated per assertion below.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_package
from golden_util import GOLDEN, config_cube, mulberry32_at, dec_num, dec_store, default_of, expected_typed, is_default_typed, load_cases, same_typed
from oracle.oracle import OracleStore, to_typed

pytestmark = pytest.mark.gpu

pkg = load_package()
capi = pkg.capi
KAT = load_cases("store_kat.json")
RND = load_cases("store_random.json")
CFG = load_cases("configs.json")


def dense_input(case, key="in", default_key="default"):
    """Golden store dump -> float64 dense array with NaN/0 default in unset cells, ascending."""
    size, keys, vals = dec_store(case[key])
    d = default_of(case, default_key)
    dense = np.full(size, d, dtype=np.float64)
    dense[keys.astype(np.int64)] = vals
    ascending = bool(np.all(np.diff(keys.astype(np.int64)) > 0)) if len(keys) > 1 else True
    return dense, ascending


def both_stores(case, key="in", default_key="default"):
    dense, ascending = dense_input(case, key, default_key)
    d = default_of(case, default_key)
    # the typed store holds TypedArray-converted cells; the oracle gets the same cells
    typed = to_typed(dense, case["type"]).astype(np.float64)
    if case["type"] in ("int32", "uint32") and d != d:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o = OracleStore(len(dense), case["type"], d)
    o.set_data(typed)
    g = pkg.HipStore(len(dense), case["type"], d)
    g.set_data_f64(dense)
    return o, g, ascending


def run_both(case):
    o, g, ascending = both_stores(case)
    op = case["op"]
    if op == "drillUp":
        args = (case["oldLen"], case["newLen"], case["maps"], case["method"])
        return o.drill_up(*args), g.drill_up(*args), ascending
    if op == "drillDown":
        dist = [dec_num(x) for x in case["distributions"]] if case.get("distributions") is not None else None
        args = (case["oldLen"], case["newLen"], case["maps"], case["method"], dist)
        return o.drill_down(*args), g.drill_down(*args), ascending
    if op == "dice":
        args = (case["oldLen"], case["newLen"], case["sel"])
        return o.dice(*args), g.dice(*args), ascending
    if op == "reorder":
        args = (case["oldLen"], case["perm"])
        return o.reorder(*args), g.reorder(*args), ascending
    if op == "load":
        ho, hg, _ = both_stores(case, "his", "hisDefault")
        o.load(ho, case["myLen"], case["hisLen"], case["hisToMine"])
        g.load(hg, case["myLen"], case["hisLen"], case["hisToMine"])
        return o, g, ascending
    raise AssertionError(op)


@pytest.mark.parametrize("case", KAT + RND, ids=lambda c: c["name"])
def test_hip_matches_oracle_and_golden(case):
    if "throws" in case:
        with pytest.raises(pkg.OlapError, match=case["throws"]):
            run_both(case)
        return
    o, g, ascending = run_both(case)
    ev, es = expected_typed(o)
    gv, gs = g.get_data(), g.get_status()
    assert g.size == o.size
    assert np.array_equal(gs, es), "status mask differs from the oracle"
    assert same_typed(gv, ev), f"values differ from the oracle: {gv[:8]} vs {ev[:8]}"
    assert np.array_equal(g.keys(), np.nonzero(es)[0])
    # direct comparison with the reference's own output (no oracle in between) whenever the
    # reference's Map was filled in ascending order and the case does not hinge on values the
    # typed store cannot hold
    if ascending and case["op"] != "load":
        size, keys, vals = dec_store(case["out"])
        ref = OracleStore(size, case["type"], default_of(case))
        for k, v in zip(keys, vals):
            ref.set(int(k), float(v))
        rv, rs = expected_typed(ref)
        assert np.array_equal(gs, rs) and same_typed(gv, rv), "differs from the reference's golden output"


def test_store_accessors():
    s = pkg.HipStore(6, "float32", 0.0)
    assert s.size == 6 and s.byte_length == 24 and s.type == "float32"
    assert np.array_equal(s.get_data(), np.zeros(6, np.float32)) and s.count_set() == 0
    s.set_data(np.array([1, 2, 0, 8, 16, 32], np.float32))
    assert s.count_set() == 5 and s.total == 59
    assert s.get_value(3) == (8.0, True) and s.get_value(2) == (0.0, False)
    s.set_value(2, 4)
    s.set_value(0, None)
    assert np.array_equal(s.get_data(), np.array([0, 2, 4, 8, 16, 32], np.float32))
    assert np.array_equal(s.keys(), [1, 2, 3, 4, 5])
    c = s.clone()
    s.fill(7)
    assert np.array_equal(s.get_data(), np.full(6, 7, np.float32)) and s.count_set() == 6
    assert np.array_equal(c.get_data(), np.array([0, 2, 4, 8, 16, 32], np.float32))
    with pytest.raises(pkg.OlapError, match="value length is invalid: 6 !== 5"):
        s.set_data(np.zeros(5, np.float32))
    n = pkg.HipStore(3, "float32", float("nan"))
    assert np.all(np.isnan(n.get_data())) and n.count_set() == 0
    n.set_value(1, 0.0)
    assert n.get_value(1) == (0.0, True) and np.isnan(n.get_value(0)[0])
    u = pkg.HipStore(3, "uint32", float("nan"))
    u.set_data_f64([5, float("nan"), 0])
    assert np.array_equal(u.get_status(), [2, 0, 2])
    assert np.array_equal(u.get_data(), np.array([5, 0, 0], np.uint32))
    f = u.get_data_f64()
    assert f[0] == 5 and np.isnan(f[1]) and f[2] == 0


@pytest.mark.parametrize("case", CFG, ids=lambda c: c["name"])
def test_configs_against_reference_output(case):
    """BASELINE.json configs 1 (10^3) and 2 (10^6): device-generated cube, drillUp(sum) to 'all' on
    one axis, compared with the output of the reference itself (tests/golden/config*.f32)."""
    lens, axis = case["lens"], case["axis"]
    n = int(np.prod(lens))
    vals, keep = config_cube(n, case["seed"], case["frac"])
    s = pkg.HipStore(n, "float32", 0.0)
    # the on-device generator must produce the same cube as the golden generator
    pkg.capi.check(pkg.lib().olap_fill_seeded(s.values_ptr, s.status_ptr, n, 0, 2, case["seed"], case["frac"], None))
    pkg.capi.check(pkg.lib().olap_device_synchronize())
    assert np.array_equal(s.get_data(), vals) and np.array_equal(s.get_status() == 2, keep)
    new_len = list(lens)
    new_len[axis] = 1
    maps = [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    out = s.drill_up(lens, new_len, maps, "sum")
    assert out.count_set() == case["outKeys"]
    if "out" in case:
        ref32 = np.asarray([dec_num(x) for x in case["out"]], dtype=np.float64).astype(np.float32)
        refp = ref32 != 0
    else:
        ref32 = np.fromfile(os.path.join(GOLDEN, case["name"] + ".f32"), dtype=np.float32)
        refp = np.fromfile(os.path.join(GOLDEN, case["name"] + ".present.u8"), dtype=np.uint8).astype(bool)
    assert np.array_equal(out.get_status() == 2, refp)
    assert np.array_equal(out.get_data(), ref32)  # bit-exact vs Math.fround(reference)


def test_plan_on_raw_pointers_and_mask_paths():
    """The plan API on raw device pointers, with and without an input mask, gives the same cells."""
    lens = [7, 12, 20]
    n = int(np.prod(lens))
    vals, keep = config_cube(n, 99, 0.5)
    s = pkg.HipStore(n, "float32", 0.0)
    s.set_data(vals)
    gmap = (np.arange(12) % 5).astype(np.uint32)
    maps = [np.arange(7, dtype=np.uint32), gmap, np.arange(20, dtype=np.uint32)]
    for method in ("sum", "average", "highest", "lowest", "first", "last", "product"):
        plan = pkg.Plan.drillup("float32", 0.0, method, lens, [7, 5, 20], maps)
        a = pkg.HipStore(plan.out_cells, "float32", 0.0)
        b = pkg.HipStore(plan.out_cells, "float32", 0.0)
        plan.run(s.values_ptr, None, a.values_ptr, a.status_ptr)
        plan.run(s.values_ptr, s.status_ptr, b.values_ptr, b.status_ptr)
        pkg.capi.check(pkg.lib().olap_device_synchronize())
        assert same_typed(a.get_data(), b.get_data()) and np.array_equal(a.get_status(), b.get_status())
        o = OracleStore(n, "float32", 0.0)
        o.set_data(vals.astype(np.float64))
        ev, es = expected_typed(o.drill_up(lens, [7, 5, 20], maps, method))
        assert same_typed(a.get_data(), ev) and np.array_equal(a.get_status(), es), method


@pytest.mark.parametrize("shape,axis", [([10] * 8, 0), ([10] * 8, 4), ([10] * 8, 7), ([3652, 100, 274], 0), ([120, 100, 274], 1)])
def test_full_size_properties(shape, axis):
    """BASELINE.json's full sizes (10^8 cells): independent float64 recomputation with numpy in the
    same accumulation order (bit-exact), plus total preservation."""
    n = int(np.prod(shape))
    s = pkg.HipStore(n, "float32", 0.0)
    pkg.capi.check(pkg.lib().olap_fill_seeded(s.values_ptr, s.status_ptr, n, 0, 2, 20240807, 1.0, None))
    K = shape[axis]
    if shape[0] == 3652 and axis == 0:  # day -> month style contiguous runs of ~30
        gmap = (np.arange(K) // 31).astype(np.uint32)
    elif K == 100:
        gmap = (np.arange(K) % 10).astype(np.uint32)  # interleaved groups
    else:
        gmap = np.zeros(K, np.uint32)
    G = int(gmap.max()) + 1
    new_len = list(shape)
    new_len[axis] = G
    maps = [gmap if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(shape)]
    out = s.drill_up(shape, new_len, maps, "sum")
    got = out.get_data()
    x = s.get_data().reshape(int(np.prod(shape[:axis])), K, int(np.prod(shape[axis + 1:])))
    ref = np.zeros((x.shape[0], G, x.shape[2]), dtype=np.float64)
    for k in range(K):  # ascending k == the reference's accumulation order
        ref[:, gmap[k], :] += x[:, k, :]
    assert np.array_equal(got, ref.astype(np.float32).ravel())
    assert np.all(out.get_status() == 2)
    assert abs(out.total - s.total) <= 1e-6 * s.total


def _day_to_month():
    """time(day, 2010-01-01 .. 2019-12-31) -> month index: an independent Gregorian computation (datetime), the map
    TimeDimension.getGroupIndexFromRootIndexMap('month') yields (src/dimension/time.js:182-197)."""
    import datetime

    d0 = datetime.date(2010, 1, 1)
    days = [d0 + datetime.timedelta(days=i) for i in range(3652)]
    assert days[-1] == datetime.date(2019, 12, 31)
    return np.array([(d.year - 2010) * 12 + d.month - 1 for d in days], np.uint32)


def _numpy_rollup(x, gmap, G, method):
    """[outer, K, inner] float64 + presence -> the reference's drillUp in ascending-k order (in-memory.js:282-331):
    only set cells contribute; average divides by the number of contributions (< 65536 here)."""
    vals, pres = x
    outer, K, inner = vals.shape
    acc = np.zeros((outer, G, inner))
    cnt = np.zeros((outer, G, inner), np.int64)
    for k in range(K):
        g = gmap[k]
        v, p = vals[:, k, :], pres[:, k, :]
        if method in ("sum", "average"):
            acc[:, g, :] += np.where(p, v, 0.0)
        elif method == "first":
            acc[:, g, :] = np.where(p & (cnt[:, g, :] == 0), v, acc[:, g, :])
        elif method == "last":
            acc[:, g, :] = np.where(p, v, acc[:, g, :])
        cnt[:, g, :] += p
    if method == "average":
        acc = np.where(cnt > 0, acc / np.maximum(cnt, 1), 0.0)
    return acc, cnt > 0


def test_config5_as_written():
    """BASELINE configs[4] at full size: time(day)=3652 x location(city)=100 x sku=274 (1.0006e8 cells), four measures
    with rules sum / average / first / last; drillUp(time, month) with the REAL calendar map (runs of 28-31 days),
    then drillUp(location, country) on the result.  Expected values: numpy float64 in ascending-k order, rounded to
    Float32 after each operation as the typed store does — bit-exact (in-memory.js:282-331)."""
    lens = [3652, 100, 274]
    n = int(np.prod(lens))
    d2m = _day_to_month()
    assert d2m.max() == 119 and np.bincount(d2m).min() == 28 and np.bincount(d2m).max() == 31
    c2c = (np.arange(100) // 10).astype(np.uint32)  # 10 cities per country
    ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
    for m, method in enumerate(("sum", "average", "first", "last")):
        s = pkg.HipStore(n, "float32", 0.0)
        pkg.capi.check(pkg.lib().olap_fill_seeded(s.values_ptr, None, n, 0, 2, 20240807 + m, 0.9, None))  # 10 % of the cells unset
        months = s.drill_up(lens, [120, 100, 274], [d2m, ident(100), ident(274)], method)
        countries = months.drill_up([120, 100, 274], [120, 10, 274], [ident(120), c2c, ident(274)], method)
        x = s.get_data().astype(np.float64).reshape(1, 3652, 27400)
        assert 0.89 < np.count_nonzero(x) / n < 0.91
        ref1, set1 = _numpy_rollup((x, x != 0), d2m, 120, method)
        ref1 = ref1.astype(np.float32)  # the typed store rounds after every operation
        assert np.array_equal(months.get_data(), ref1.ravel()), method + " day->month"
        assert np.array_equal(months.get_status() == 2, (set1 & (ref1 != 0)).ravel()), method
        y = ref1.astype(np.float64).reshape(120, 100, 274)
        ref2, set2 = _numpy_rollup((y, y != 0), c2c, 10, method)
        ref2 = ref2.astype(np.float32)
        assert np.array_equal(countries.get_data(), ref2.ravel()), method + " city->country"
        assert np.array_equal(countries.get_status() == 2, (set2 & (ref2 != 0)).ravel()), method
        del s, months, countries, x, y


@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", float("nan"))])
def test_config5_chain_against_the_oracle(type_name, default):
    """The same chain (real calendar map over 2010-2011, then city -> country; sum / average / first / last) at a size
    the oracle covers, against OracleStore."""
    d2m = _day_to_month()[:730]
    lens = [730, 20, 9]
    n = int(np.prod(lens))
    c2c = (np.arange(20) // 5).astype(np.uint32)
    ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
    rng = np.random.default_rng(5)
    vals = rng.integers(-20, 21, size=n).astype(np.float64)
    vals = np.where(rng.random(n) < 0.8, vals, default)
    for method in ("sum", "average", "first", "last"):
        o = OracleStore(n, type_name, default)
        o.set_data(vals)
        g = pkg.HipStore(n, type_name, default)
        g.set_data_f64(vals)
        m1 = [d2m, ident(20), ident(9)]
        m2 = [ident(24), c2c, ident(9)]
        og = o.drill_up(lens, [24, 20, 9], m1, method)
        gg = g.drill_up(lens, [24, 20, 9], m1, method)
        ev, es = expected_typed(og)
        assert same_typed(gg.get_data(), ev) and np.array_equal(gg.get_status(), es), method
        # second step from what the typed store holds (it rounds after every operation; see test_typed_storage_differences)
        o2 = OracleStore(24 * 20 * 9, type_name, default)
        o2.set_data(np.where(es == 2, ev.astype(np.float64), default))
        ev2, es2 = expected_typed(o2.drill_up([24, 20, 9], [24, 4, 9], m2, method))
        g2 = gg.drill_up([24, 20, 9], [24, 4, 9], m2, method)
        assert same_typed(g2.get_data(), ev2) and np.array_equal(g2.get_status(), es2), method


@pytest.mark.parametrize("shape", [[10] * 9, [320, 5, 5, 5, 5, 5, 5, 10, 20]], ids=["literal", "friendly"])
def test_config4_shapes_on_one_gpu(shape):
    """BASELINE configs[3]'s 10^9-cell cube, both shapes of SURVEY 8(e), whole on ONE GPU: drillUp(sum) of dimension 0
    -> all against float64 column sums on slices of the output (the N = 1 point of the sharded series)."""
    n = int(np.prod(shape))
    n_out = n // shape[0]
    s = pkg.HipStore(n, "float32", 0.0)
    pkg.capi.check(pkg.lib().olap_fill_seeded(s.values_ptr, None, n, 0, 2, 20240807, 1.0, None))
    maps = [np.zeros(shape[0], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in shape[1:]]
    out = s.drill_up(shape, [1] + shape[1:], maps, "sum")
    got = out.get_data()
    assert got.size == n_out
    # the generator is position-addressed: recompute three slices of every row on the host
    for lo in (0, n_out // 2 - 500, n_out - 1000):
        cols = np.zeros(1000)
        for r in range(shape[0]):
            u = mulberry32_at(20240807, 2 * (np.arange(r * n_out + lo, r * n_out + lo + 1000, dtype=np.uint64)) + 1)
            cols += (0.5 + u).astype(np.float32).astype(np.float64)
        assert np.array_equal(got[lo:lo + 1000], cols.astype(np.float32)), (shape[0], lo)
    assert abs(out.total - s.total) <= 1e-6 * s.total


@pytest.mark.parametrize("seed", range(154))
def test_fused_dice_drillup_equals_two_steps(seed):
    """olap_dice_drillup_plan == dice then drillUp of the oracle (selection with reordering, unknown
    items and duplicates; one rolled-up dimension; every method, type and default)."""
    rng = np.random.default_rng(1000 + seed)
    shapes = [[6, 5, 8], [12], [3, 4, 5, 4], [7, 16], [5, 3, 64], [4, 6, 512], [3, 40, 129], [2, 3, 4096], [300, 8], [9, 1000], [2, 5, 3, 260]]
    old_len = shapes[seed % len(shapes)]
    nd = len(old_len)
    type_name = ["float32", "float64", "int32", "uint32"][seed % 4]
    default = float("nan") if seed % 3 == 0 else 0.0
    method = ["sum", "average", "highest", "lowest", "first", "last", "product"][seed % 7]
    n = int(np.prod(old_len))
    vals = rng.integers(1, 9, size=n).astype(np.float64) * (1 if type_name.endswith("int32") else 0.5)
    unset = rng.random(n) < 0.3
    dense = np.where(unset, default, vals)
    sel, mid_len = [], []
    for l in old_len:
        if rng.random() < 0.3:
            s = np.arange(l)
        else:
            s = rng.permutation(l)[: max(1, int(l * 0.7))]
            if rng.random() < 0.5:
                s = np.sort(s)
            if rng.random() < 0.3:
                s = np.insert(s, rng.integers(0, len(s) + 1), -1)
            if rng.random() < 0.2 and len(s) > 1:
                s = np.append(s, s[0])  # duplicate: only the last occurrence receives the cells
        sel.append(s.astype(np.int32))
        mid_len.append(len(s))
    axis = int(rng.integers(0, nd))
    groups = max(1, mid_len[axis] // 2)
    labels = rng.integers(0, groups, size=mid_len[axis])
    # first-appearance numbering like GenericDimension.addAttribute
    seen, amap = {}, []
    for x in labels:
        seen.setdefault(int(x), len(seen))
        amap.append(seen[int(x)])
    maps = [np.arange(m, dtype=np.uint32) for m in mid_len]
    maps[axis] = np.asarray(amap, dtype=np.uint32)
    new_len = list(mid_len)
    new_len[axis] = len(seen)

    o = OracleStore(n, type_name, default)
    o.set_data(dense)
    diced = o.dice(old_len, mid_len, sel)
    # A reordering dice leaves the reference's Map in SOURCE insertion order, which first/last then
    # follow; a dense buffer has only cell order, so the expectation is the oracle on
    # the diced cells re-inserted ascending (what chaining the two dense operations gives).
    dv, dp = diced.dense()
    redense = OracleStore(diced.size, type_name, default)
    for i in np.nonzero(dp)[0]:
        redense.set(int(i), float(dv[i]))
    ev, es = expected_typed(redense.drill_up(mid_len, new_len, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.dice_drillup(old_len, mid_len, new_len, sel, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)
    two = g.dice(old_len, mid_len, sel).drill_up(mid_len, new_len, maps, method)
    assert same_typed(two.get_data(), out.get_data()) and np.array_equal(two.get_status(), out.get_status())


def test_config3_chain_fused_vs_unfused():
    """BASELINE config 3: slice(dimension1,item3) -> dice(dimension4,[1,4,7]) -> drillUp(dimension0,all)
    on the 10^8-cell cube, fused and unfused, against numpy float64 in the same order."""
    shape = [10] * 8
    n = 10 ** 8
    s = pkg.HipStore(n, "float32", 0.0)
    pkg.capi.check(pkg.lib().olap_fill_seeded(s.values_ptr, None, n, 0, 2, 20240807, 1.0, None))
    ident = [np.arange(10, dtype=np.int32) for _ in range(8)]
    # slice = dice to one item + drillUp of that dimension to 'all'
    sel1 = list(ident)
    sel1[1] = np.array([3], np.int32)
    mid1 = [10, 1, 10, 10, 10, 10, 10, 10]
    umap = lambda lens, axis: [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    a = s.dice_drillup(shape, mid1, mid1, sel1, umap(mid1, 1), "sum")  # [10,1,10,...]: dim1 already 'all'
    lens2 = mid1
    sel2 = [np.arange(l, dtype=np.int32) for l in lens2]
    sel2[4] = np.array([1, 4, 7], np.int32)
    mid2 = list(lens2)
    mid2[4] = 3
    new2 = list(mid2)
    new2[0] = 1
    fused = a.dice_drillup(lens2, mid2, new2, sel2, umap(mid2, 0), "sum")
    unfused = s.dice(shape, mid1, sel1).drill_up(mid1, mid1, umap(mid1, 1), "sum").dice(lens2, mid2, sel2).drill_up(mid2, new2, umap(mid2, 0), "sum")
    x = s.get_data().reshape(shape)[:, 3][:, :, :, [1, 4, 7]]  # dims: 0,2,3,4',5,6,7
    ref = np.zeros(x.shape[1:], dtype=np.float64)
    for k in range(10):
        ref += x[k]
    assert np.array_equal(fused.get_data(), ref.astype(np.float32).ravel())
    assert np.array_equal(unfused.get_data(), fused.get_data())


@pytest.mark.parametrize("method", ["sum", "average", "highest", "lowest", "first", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)])
@pytest.mark.parametrize("lens,groups", [([7, 1500, 3], 4), ([3, 3652, 10], 10), ([5, 5200, 1], 5)])
def test_group_tile_long_groups_share_output_cells(method, type_name, default, lens, groups, monkeypatch):
    """Contiguous LONG groups (>= 256 members: day -> year) over short row pieces: a tile of the group-tile regime
    holds one or two groups, i.e. a handful of output cells, and L lanes share each of them — a plain running sum +
    shuffle for sum / average over a 0 default, the reduce regime's Partial merge (in member order) for everything
    else.  Picks are exact; float64 sums are re-associated (the inputs are quarter-integers: still exact here)."""
    monkeypatch.setenv("OLAP_REDUCE_MAX_CELLS", "0")  # (below 131 072 output cells the few-outputs reduce regime would take these small cubes)
    rng = np.random.default_rng(sum(lens) + len(method))
    n = int(np.prod(lens))
    K = lens[1]
    bounds = np.linspace(0, K, groups + 1).astype(int)
    amap = np.repeat(np.arange(groups), np.diff(bounds)).astype(np.uint32)
    assert np.diff(bounds).min() >= 256
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) * np.where(rng.random(n) < 0.01, 2.0, 1.0)
        if type_name == "uint32":
            vals = np.abs(vals)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -50, 51, size=n).astype(np.float64)
        if type_name.startswith("float"):
            vals = vals * 0.25
    dense = np.where(rng.random(n) < 0.3, default, vals)
    new = [lens[0], groups, lens[2]]
    maps = [np.arange(lens[0], dtype=np.uint32), amap, np.arange(lens[2], dtype=np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    item = 8 if type_name == "float64" else 4
    fits = int(np.diff(bounds).max()) * lens[2] <= 16384 // item - 16 // item  # a whole group per tile
    assert ("gtile" if fits else "flat") in plan.kernel_name, plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    gv = out.get_data()
    if type_name == "float64" and method in ("sum", "average", "product"):
        assert np.allclose(gv, ev, rtol=1e-12, atol=0, equal_nan=True)
    else:
        assert same_typed(gv, ev)


@pytest.mark.parametrize("method", ["sum", "average", "highest", "lowest", "first", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)])
def test_split_regime_few_outputs_long_groups(method, type_name, default):
    """[6000, 7] -> [2, 7]: 14 output cells, groups of ~3000 rows: the reduce regime (cooperative
    segments + merge).  Picks are exact; float64 sums are re-associated (1e-12 relative; the inputs
    here are quarter-integers, so float32/int results are still exact)."""
    rng = np.random.default_rng(5)
    lens = [6000, 7]
    n = 42000
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) * np.where(rng.random(n) < 0.01, 2.0, 1.0)
    else:
        vals = rng.integers(-50, 51, size=n).astype(np.float64)
        if type_name == "uint32":
            vals = np.abs(vals)
        if type_name.startswith("float"):
            vals = vals * 0.25
    unset = rng.random(n) < 0.3
    dense = np.where(unset, default, vals)
    row_map = (np.arange(6000) % 2).astype(np.uint32)  # interleaved groups -> `order` table in use
    maps = [row_map, np.arange(7, dtype=np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, [2, 7], maps)
    assert "reduce" in plan.kernel_name or "split" in plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, [2, 7], maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, [2, 7], maps, method)
    assert np.array_equal(out.get_status(), es)
    gv = out.get_data()
    if type_name == "float64" and method in ("sum", "average", "product"):
        assert np.allclose(gv, ev, rtol=1e-12, atol=0, equal_nan=True)
    else:
        assert same_typed(gv, ev)


@pytest.mark.parametrize("lens,axis", [([300, 4000], 1), ([5, 70000], 1), ([40000, 200], 0), ([1, 100000], 1), ([100000], 0)])
@pytest.mark.parametrize("method", ["sum", "average", "first", "last", "highest", "product"])
def test_reduce_regime_shapes(lens, axis, method):
    """Long groups with few output cells in every geometry of the reduce regime (cooperative
    workgroups for inner <= 128, lane-split for wider rows, one- and multi-segment)."""
    rng = np.random.default_rng(11)
    n = int(np.prod(lens))
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0)
    else:
        vals = rng.integers(-8, 9, size=n).astype(np.float64) * 0.5
    dense = np.where(rng.random(n) < 0.4, 0.0, vals)
    K = lens[axis]
    amap = (np.arange(K) * 3 // K).astype(np.uint32) if K >= 3 else np.zeros(K, np.uint32)  # 3 contiguous groups
    new = list(lens)
    new[axis] = int(amap.max()) + 1
    maps = [amap if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    plan = pkg.Plan.drillup("float32", 0.0, method, lens, new, maps)
    assert "reduce" in plan.kernel_name or "split" in plan.kernel_name, plan.kernel_name
    o = OracleStore(n, "float32", 0.0)
    o.set_data(dense)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, "float32", 0.0)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("lens,axis", [([5000, 300], 1), ([4200, 260, 4], 1), ([1000, 264, 8], 1), ([3, 40000, 2], 1), ([100000], 0),
                                       ([2000000], 0), ([400000, 10], 0), ([2, 40000, 64], 1), ([70000, 128], 0), ([3, 1000, 12], 1),
                                       ([7, 4097, 100], 1), ([2, 9000, 68], 1), ([1, 50000, 24], 1),
                                       # inner = 1 with rows at odd offsets (K % 4 != 0): aligned groups, masked ends
                                       ([5000, 301], 1), ([3, 40001], 1), ([100001], 0), ([130000, 257], 1)])
@pytest.mark.parametrize("method", ["sum", "average", "first", "last", "highest", "lowest", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("uint32", float("nan")), ("float64", 0.0), ("float64", float("nan"))])
def test_reduce_regime_to_all(lens, axis, method, type_name, default):
    """'-> all' roll-ups of contiguous rows with few output cells: the 16-byte cooperative form
    (drillup_reduce4_kernel: four 4-byte cells or two 8-byte cells per lane) in each of its geometries — wave-shuffle row
    merge (inner 1, 2, 4, 8, 12, 64) and LDS tree (inner 10, 128), one segment per group (result written by the reduction
    itself), a few segments (lane-per-cell merge) and hundreds (wave-per-cell merge), with and without the status mask."""
    rng = np.random.default_rng(17)
    n = int(np.prod(lens))
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) if type_name.startswith("float") else np.ones(n)
        vals = vals * np.where(rng.random(n) < 4.0 / lens[axis], 2.0, 1.0)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=n).astype(np.float64)
        if type_name.startswith("float"):
            vals = vals * 0.5
    dense = np.where(rng.random(n) < 0.4, default, vals)
    new = list(lens)
    new[axis] = 1
    maps = [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    assert "reduce4" in plan.kernel_name, plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name == "uint32":
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("lens,sizes", [
    ([7, 3652, 30], "calendar"),        # day -> month in the middle of the cube: rows of 109 560 cells
    ([3, 3653, 5], "calendar"),         # cell count not a multiple of 4: the last tile ends at the buffer's end
    ([5, 100000, 1], "thirty"),         # inner = 1, 3 334 groups: several tiles of up to 1 024 groups
    ([2, 9000, 3], "ragged"),           # groups of 1..40 members, odd offsets
    ([4, 5000, 127], "ragged30"),       # widest inner of the regime (groups of <= 30 members still fit a tile)
])
@pytest.mark.parametrize("method", ["sum", "average", "highest", "first", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float64", float("nan")), ("uint32", float("nan"))])
def test_group_tile_regime(lens, sizes, method, type_name, default):
    """Contiguous groups whose rows do not fit LDS (drillup_gtile_kernel): tiles of whole groups."""
    rng = np.random.default_rng(31)
    K = lens[1]
    if sizes == "calendar":
        days = np.arange(np.datetime64("2010-01-01"), np.datetime64("2010-01-01") + K)
        months = days.astype("datetime64[M]").astype(np.int64)
        amap = (months - months[0]).astype(np.uint32)
    elif sizes == "thirty":
        amap = (np.arange(K) // 30).astype(np.uint32)
    else:
        amap = np.repeat(np.arange(K), rng.integers(1, 31 if sizes == "ragged30" else 41, size=K))[:K].astype(np.uint32)
        amap = np.unique(amap, return_inverse=True)[1].astype(np.uint32)
    n = int(np.prod(lens))
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) if type_name != "uint32" else np.ones(n)
        vals = vals * np.where(rng.random(n) < 0.05, 2.0, 1.0)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=n).astype(np.float64)
        if type_name != "uint32":
            vals = vals * 0.5
    dense = np.where(rng.random(n) < 0.4, default, vals)
    new = [lens[0], int(amap.max()) + 1, lens[2]]
    maps = [np.arange(lens[0], dtype=np.uint32), amap, np.arange(lens[2], dtype=np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    # the plan takes the tile form when every group fits a 16 KiB tile and a tile keeps >= 64 lanes busy
    assert plan.kernel_name in ("drillup_gtile_kernel", "drillup_flat_kernel"), plan.kernel_name
    if type_name == "float32" and lens in ([7, 3652, 30], [5, 100000, 1], [2, 9000, 3]):
        assert plan.kernel_name == "drillup_gtile_kernel"
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name == "uint32":
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("lens,kind", [
    ([5, 1000, 100], "mod100"),      # 100 interleaved groups of 10 members, pieces of 100 cells
    ([3, 900, 12], "mod50"),         # 18 members x 12 cells per group
    ([2, 640, 64], "random"),        # a random map: ragged group sizes, members anywhere
    ([4, 3000, 8], "long"),          # groups of 300 members
    ([3, 700, 6], "mod35"),          # pieces of 6 cells: 8-byte lanes
])
@pytest.mark.parametrize("method", ["sum", "average", "highest", "first", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float64", float("nan")), ("uint32", float("nan"))])
def test_flat_regime_interleaved_groups(lens, kind, method, type_name, default, monkeypatch):
    """Interleaved groups over short row pieces whose rows do not fit LDS: the flat form (a lane per 16 bytes of output walks
    its group's member list).  (Gathering a tile's members into LDS and reducing there, as for contiguous groups, was
    measured: 103 us against the flat form's 81 us on [1000,1000,100] -> 100 groups, modular, blocked or random maps.)"""
    monkeypatch.setenv("OLAP_REDUCE_MAX_CELLS", "0")  # (small cubes: keep the few-outputs regime out of the way)
    rng = np.random.default_rng(77)
    K = lens[1]
    if kind.startswith("mod"):
        amap = (np.arange(K) % int(kind[3:])).astype(np.uint32)
    elif kind == "long":
        amap = (np.arange(K) % 10).astype(np.uint32)
    else:
        amap = rng.integers(0, 40, size=K).astype(np.uint32)
        amap = np.unique(amap, return_inverse=True)[1].astype(np.uint32)
    n = int(np.prod(lens))
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) if type_name != "uint32" else np.ones(n)
        vals = vals * np.where(rng.random(n) < 0.05, 2.0, 1.0)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=n).astype(np.float64)
        if type_name != "uint32":
            vals = vals * 0.5
    dense = np.where(rng.random(n) < 0.4, default, vals)
    new = [lens[0], int(amap.max()) + 1, lens[2]]
    maps = [np.arange(lens[0], dtype=np.uint32), amap, np.arange(lens[2], dtype=np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    assert plan.kernel_name == "drillup_flat_kernel", plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name == "uint32":
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("lens", [[37, 5, 300], [9, 271], [3, 4, 1000]])
@pytest.mark.parametrize("method", ["sum", "average", "highest", "lowest", "first", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("float64", 0.0), ("int32", 0.0), ("uint32", float("nan"))])
def test_tile_regime_long_rows_every_rule(lens, method, type_name, default, monkeypatch):
    """The LAST dimension (256 - 4 096 items) rolled up to 'all' with more rows than the few-outputs regime takes: a tile
    holds a dozen rows, 16 lanes share each — plain float64 sums for sum / average over a 0 default, the Partial state
    machine merged lane to lane for every other rule, the mask and the NaN default."""
    monkeypatch.setenv("OLAP_REDUCE_MAX_CELLS", "0")
    rng = np.random.default_rng(len(lens) * 100 + lens[-1])
    n = int(np.prod(lens))
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) if type_name not in ("uint32",) else np.ones(n)
        vals = vals * np.where(rng.random(n) < 0.02, 2.0, 1.0)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -9, 10, size=n).astype(np.float64)
        if type_name.startswith("float"):
            vals = vals * 0.5
    dense = np.where(rng.random(n) < 0.5, default, vals)
    dense.reshape(-1, lens[-1])[0, :] = default  # a row nobody set
    new = lens[:-1] + [1]
    maps = [np.arange(l, dtype=np.uint32) for l in lens[:-1]] + [np.zeros(lens[-1], np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    assert plan.kernel_name == "drillup_tile_kernel", plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)  # (half-integers: the re-associated float64 sums are exact)


@pytest.mark.parametrize("default", [0.0, float("nan")])
@pytest.mark.parametrize("lens,groups", [([64, 3], 1), ([2, 6000, 1], 2000), ([5, 9, 4], 3)])
def test_product_that_hits_the_default_restarts(lens, groups, default):
    """`product` runs as a plain chain first in the LDS regimes; a running product that equals the default on the way — an
    underflow to 0 under the 0 default, inf x 0 = NaN under the NaN default — drops the key and restarts with the next
    member (in-memory.js:311-318): those groups are walked again through the exact state machine.  Against the oracle."""
    rng = np.random.default_rng(3)
    axis = 1 if len(lens) > 1 else 0
    K = lens[axis]
    n = int(np.prod(lens))
    vals = rng.choice([2.0, -3.0, 0.5, 7.0], size=n)
    cube = vals.reshape(lens[0], K, -1)
    per = K // groups
    for o in range(cube.shape[0]):
        for g in range(0, groups, 2):  # every other group: the first two members wipe the running product out
            k0 = g * per
            if default == 0.0:
                cube[o, k0, :] = 1e-200
                cube[o, k0 + 1, :] = 1e-200
            else:
                cube[o, k0, :] = np.inf
                cube[o, k0 + 1, :] = 0.0
    dense = cube.ravel().copy()
    dense[rng.random(n) < 0.1] = default
    amap = (np.arange(K) // per).astype(np.uint32)
    new = list(lens)
    new[axis] = groups
    maps = [amap if d == axis else np.arange(l, dtype=np.uint32) for d, l in enumerate(lens)]
    plan = pkg.Plan.drillup("float64", default, "product", lens, new, maps)
    assert plan.kernel_name in ("drillup_tile_kernel", "drillup_gtile_kernel"), plan.kernel_name
    o = OracleStore(n, "float64", default)
    o.set_data(dense)
    ev, es = expected_typed(o.drill_up(lens, new, maps, "product"))
    g = pkg.HipStore(n, "float64", default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, "product")
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)
    assert np.isfinite(ev[es == 2]).mean() > 0.9  # (the wiped-out groups restarted: their products are ordinary numbers)


@pytest.mark.parametrize("type_name,default,flag", [("float32", 0.0, False), ("int32", 0.0, False), ("float64", float("nan"), True), ("uint32", float("nan"), False)])
def test_drilldown_plans_are_reused_across_stores(type_name, default, flag):
    """drillDown plans without distributions are cached by (cell type, default, rule, shapes, maps): three stores with
    different cells take the same plan one after the other, each against the oracle (the integer remainder rule included,
    also for a declared-integer measure in float64 cells)."""
    rng = np.random.default_rng(11)
    old_len, new_len = [4, 6, 130], [4, 15, 130]
    child = np.repeat(np.arange(6), [1, 4, 2, 3, 2, 3]).astype(np.uint32)
    maps = [np.arange(4, dtype=np.uint32), child, np.arange(130, dtype=np.uint32)]
    n = int(np.prod(old_len))
    for round_ in range(3):
        vals = rng.integers(0 if type_name == "uint32" else -50, 200, size=n).astype(np.float64)
        dense = np.where(rng.random(n) < 0.3, default, vals)
        declared = "int32" if flag else type_name
        o = OracleStore(n, declared, default)
        typed = dense if flag else to_typed(dense, type_name).astype(np.float64)
        if type_name in ("int32", "uint32") and default != default:
            typed = np.where(np.isnan(dense), np.nan, typed)
        o.set_data(typed)
        g = pkg.HipStore(n, type_name, default)
        g.set_data_f64(dense)
        for method in ("sum", "average"):
            want = o.drill_down(old_len, new_len, maps, method)
            out = g.drill_down(old_len, new_len, maps, method, integer_measure=flag)
            if flag:
                v, present = want.dense()
                assert np.array_equal(out.get_data_f64(), np.where(present, v, default), equal_nan=True), (round_, method)
            else:
                ev, es = expected_typed(want)
                assert np.array_equal(out.get_status(), es), (round_, method)
                assert same_typed(out.get_data(), ev), (round_, method)


def _boundary_cases():
    """Seeded sample of one-axis drillUps whose extents sit on the boundaries between the kernel
    regimes (vector width, 128 vector slots, the 16 KiB tile, 256-member groups, 131 072 outputs)."""
    rng = np.random.default_rng(20240807)
    inners = [1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 100, 127, 128, 129, 255, 256, 257, 508, 511, 512, 513, 515, 516, 1000, 1001, 1023,
              1024, 1026, 4096, 4099, 5000]
    ks = [1, 2, 3, 10, 30, 255, 256, 257, 1000, 4097]
    outers = [1, 2, 3, 7, 64, 100]
    cases = []
    while len(cases) < 900:
        inner, K, outer = int(rng.choice(inners)), int(rng.choice(ks)), int(rng.choice(outers))
        if outer * K * inner > 3_000_000:
            continue
        kind = str(rng.choice(["all", "contiguous", "interleaved", "random", "identity"]))
        method = str(rng.choice(["sum", "average", "highest", "lowest", "first", "last", "product"]))
        type_name, default = [("float32", 0.0), ("float32", float("nan")), ("float64", 0.0), ("int32", 0.0), ("uint32", float("nan"))][int(rng.integers(0, 5))]
        cases.append((outer, K, inner, kind, method, type_name, default, int(rng.integers(0, 2 ** 31))))
    return cases


@pytest.mark.parametrize("outer,K,inner,kind,method,type_name,default,seed", _boundary_cases())
def test_drillup_regime_boundaries(outer, K, inner, kind, method, type_name, default, seed):
    rng = np.random.default_rng(seed)
    if kind == "all":
        amap = np.zeros(K, np.uint32)
    elif kind == "identity":
        amap = np.arange(K, dtype=np.uint32)
    elif kind == "contiguous":
        amap = np.unique(np.sort(rng.integers(0, max(1, K // 3) + 1, size=K)), return_inverse=True)[1].astype(np.uint32)
    elif kind == "interleaved":
        amap = (np.arange(K) % max(1, min(K, int(rng.integers(1, 12))))).astype(np.uint32)
    else:
        amap = np.unique(rng.integers(0, max(1, K // 2) + 1, size=K), return_inverse=True)[1].astype(np.uint32)
        # group ids must be numbered by first appearance, as GenericDimension.addAttribute does (generic.js:100-107)
        first = {}
        amap = np.array([first.setdefault(int(g), len(first)) for g in amap], dtype=np.uint32)
    lens = [outer, K, inner]
    new = [outer, int(amap.max()) + 1, inner]
    n = outer * K * inner
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) if type_name not in ("uint32",) else np.ones(n)
        vals = vals * np.where(rng.random(n) < 2.0 / max(K, 2), 2.0, 1.0)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=n).astype(np.float64)
        if type_name.startswith("float"):
            vals = vals * 0.5
    dense = np.where(rng.random(n) < 0.35, default, vals)
    maps = [np.arange(outer, dtype=np.uint32), amap, np.arange(inner, dtype=np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es), plan.kernel_name
    assert same_typed(out.get_data(), ev), plan.kernel_name


def _down_cases():
    rng = np.random.default_rng(8072024)
    inners = [1, 3, 4, 17, 32, 100, 127, 128, 129, 511, 512, 513, 515, 516, 520, 1000, 1001, 1023, 1024, 1026, 2049, 2056]
    cases = []
    while len(cases) < 260:
        inner, G, outer = int(rng.choice(inners)), int(rng.choice([1, 2, 5, 12, 40])), int(rng.choice([1, 2, 3, 9]))
        fan = int(rng.choice([1, 2, 3, 7, 19, 31]))
        if outer * G * fan * inner > 2_000_000:
            continue
        kind = str(rng.choice(["contiguous", "interleaved", "uneven"]))
        method = str(rng.choice(["sum", "average"]))
        type_name, default = [("float32", 0.0), ("float32", float("nan")), ("float64", 0.0), ("int32", 0.0), ("uint32", float("nan"))][int(rng.integers(0, 5))]
        cases.append((outer, G, fan, inner, kind, method, type_name, default, int(rng.integers(0, 2 ** 31))))
    return cases


@pytest.mark.parametrize("outer,G,fan,inner,kind,method,type_name,default,seed", _down_cases())
def test_drilldown_regime_boundaries(outer, G, fan, inner, kind, method, type_name, default, seed):
    """One refined dimension across the drillDown forms (row form, line-aligned windows, scale +
    broadcast, one lane per child) and their extent boundaries."""
    rng = np.random.default_rng(seed)
    if kind == "contiguous":
        child = np.repeat(np.arange(G), fan)
    elif kind == "interleaved":
        child = np.arange(G * fan) % G
    else:
        child = np.repeat(np.arange(G), rng.integers(1, fan + 1, size=G))
    child = child.astype(np.uint32)
    K = len(child)
    old_len, new_len = [outer, G, inner], [outer, K, inner]
    n_old = outer * G * inner
    vals = rng.integers(0 if type_name == "uint32" else -60, 120, size=n_old).astype(np.float64)
    if type_name.startswith("float"):
        vals = vals * 0.25
    dense = np.where(rng.random(n_old) < 0.3, default, vals)
    maps = [np.arange(outer, dtype=np.uint32), child, np.arange(inner, dtype=np.uint32)]
    plan = pkg.Plan.drilldown(type_name, default, method, old_len, new_len, maps)
    o = OracleStore(n_old, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_down(old_len, new_len, maps, method))
    g = pkg.HipStore(n_old, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_down(old_len, new_len, maps, method)
    assert np.array_equal(out.get_status(), es), plan.kernel_name
    assert same_typed(out.get_data(), ev), plan.kernel_name


def _dice_cases():
    rng = np.random.default_rng(772024)
    cases = []
    while len(cases) < 120:
        lens = [int(rng.choice([1, 2, 3, 7, 10, 33])), int(rng.choice([1, 4, 9, 30])), int(rng.choice([1, 3, 4, 16, 100, 128, 129, 1000]))]
        if int(np.prod(lens)) > 1_500_000:
            continue
        type_name, default = [("float32", 0.0), ("float64", float("nan")), ("int32", 0.0), ("uint32", float("nan"))][int(rng.integers(0, 4))]
        cases.append((lens, type_name, default, int(rng.integers(0, 2 ** 31))))
    return cases


@pytest.mark.parametrize("lens,type_name,default,seed", _dice_cases())
def test_dice_selections_randomized(lens, type_name, default, seed):
    """dice (in-memory.js:213-263) with random selections on every dimension: subsets, reorderings,
    unknown items (-1) and duplicates (only the last occurrence receives the cells)."""
    rng = np.random.default_rng(seed)
    sel = []
    for l in lens:
        mode = int(rng.integers(0, 4))
        if mode == 0:
            s_ = np.arange(l)
        elif mode == 1:
            s_ = np.sort(rng.choice(l, size=int(rng.integers(0, l + 1)), replace=False))
        elif mode == 2:
            s_ = rng.permutation(l)[: int(rng.integers(1, l + 1))]
        else:
            s_ = rng.integers(-1, l, size=int(rng.integers(1, l + 3)))
        sel.append(s_.astype(np.int32))
    new_len = [len(x) for x in sel]
    n = int(np.prod(lens))
    vals = rng.integers(1, 500, size=n).astype(np.float64)
    dense = np.where(rng.random(n) < 0.3, default, vals)
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.dice(lens, new_len, sel))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.dice(lens, new_len, sel)
    assert out.size == int(np.prod(new_len))
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("seed", range(60))
def test_reorder_random_permutations(seed):
    rng = np.random.default_rng(4000 + seed)
    nd = int(rng.integers(2, 6))
    lens = [int(rng.choice([1, 2, 3, 4, 7, 8, 10, 16, 25, 64, 100])) for _ in range(nd)]
    while int(np.prod(lens)) > 2_000_000:
        lens[int(rng.integers(0, nd))] = 2
    perm = [int(x) for x in rng.permutation(nd)]
    type_name, default = [("float32", 0.0), ("int32", 0.0), ("uint32", float("nan")), ("float64", float("nan"))][seed % 4]
    n = int(np.prod(lens))
    vals = rng.integers(1, 1000, size=n).astype(np.float64)
    unset = rng.random(n) < 0.3
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(np.where(unset, default, vals))
    out = g.reorder(lens, perm)
    moved = lambda a: np.ascontiguousarray(a.reshape(lens).transpose(perm)).ravel()  # noqa: E731
    assert np.array_equal(out.get_status(), moved(g.get_status()))
    assert same_typed(out.get_data(), moved(g.get_data()))


@pytest.mark.parametrize("seed", range(80))
def test_drillup_on_several_dimensions_randomized(seed):
    """Maps on more than one dimension at once (in-memory.js:270-274; drillup_generic_kernel)."""
    rng = np.random.default_rng(6000 + seed)
    nd = int(rng.integers(2, 5))
    lens = [int(rng.choice([1, 2, 3, 5, 8, 12, 30, 64])) for _ in range(nd)]
    while int(np.prod(lens)) > 400_000:
        lens[int(rng.integers(0, nd))] = 2
    maps, new = [], []
    for l in lens:
        if rng.random() < 0.35:
            m_ = np.arange(l)
        else:
            labels = rng.integers(0, max(1, l // 2) + 1, size=l)
            first = {}
            m_ = np.array([first.setdefault(int(x), len(first)) for x in labels])
        maps.append(m_.astype(np.uint32))
        new.append(int(m_.max()) + 1)
    method = ["sum", "average", "highest", "lowest", "first", "last", "product"][seed % 7]
    type_name, default = [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)][seed % 5]
    n = int(np.prod(lens))
    vals = rng.integers(0 if type_name == "uint32" else -5, 6, size=n).astype(np.float64)
    if method == "product":
        vals = np.where(vals == 0, 1.0, np.sign(vals)) if type_name != "uint32" else np.ones(n)
    dense = np.where(rng.random(n) < 0.3, default, vals)
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("seed", range(60))
def test_load_randomized(seed):
    """load (in-memory.js:139-176): another store's cells written at remapped positions; items this
    store does not have (-1) are skipped, unset cells of the other store leave ours alone."""
    rng = np.random.default_rng(7000 + seed)
    nd = int(rng.integers(1, 4))
    my_len = [int(rng.choice([1, 2, 3, 7, 20, 130])) for _ in range(nd)]
    his_len = [int(rng.choice([1, 2, 4, 9, 25, 100])) for _ in range(nd)]
    maps = []
    for ml, hl in zip(my_len, his_len):
        m_ = rng.permutation(max(ml, hl))[:hl]
        m_ = np.where(m_ < ml, m_, -1)
        maps.append(m_.astype(np.int32))
    type_name = ["float32", "int32", "uint32", "float64"][seed % 4]
    my_default = float("nan") if seed % 2 else 0.0
    his_default = float("nan") if seed % 3 == 0 else 0.0
    n_my, n_his = int(np.prod(my_len)), int(np.prod(his_len))
    mine = np.where(rng.random(n_my) < 0.5, my_default, rng.integers(1, 50, size=n_my).astype(np.float64))
    his = np.where(rng.random(n_his) < 0.4, his_default, rng.integers(50, 99, size=n_his).astype(np.float64))

    def both(n, dflt, dense):
        o = OracleStore(n, type_name, dflt)
        typed = to_typed(dense, type_name).astype(np.float64)
        if type_name in ("int32", "uint32") and dflt != dflt:
            typed = np.where(np.isnan(dense), np.nan, typed)
        o.set_data(typed)
        g = pkg.HipStore(n, type_name, dflt)
        g.set_data_f64(dense)
        return o, g

    om, gm = both(n_my, my_default, mine)
    oh, gh = both(n_his, his_default, his)
    om.load(oh, my_len, his_len, maps)
    gm.load(gh, my_len, his_len, maps)
    ev, es = expected_typed(om)
    assert np.array_equal(gm.get_status(), es)
    assert same_typed(gm.get_data(), ev)


@pytest.mark.parametrize("case", range(16))
def test_load_lane_forms(case):
    """The two wide forms of load: 16-byte lanes when the trailing dimensions are left alone (contiguous on both sides),
    and 16 bytes of the OTHER store per lane with one index decode when the innermost dimension itself is remapped
    (rows that end inside a lane's run, a cube that ends inside one, items this store lacks, 8-byte cells)."""
    rng = np.random.default_rng(7700 + case)
    shapes = [([6, 8, 12], [4, 8, 12], "outer"), ([3, 40], [5, 40], "outer"), ([9, 4, 4], [9, 4, 4], "none"), ([5, 7], [5, 7], "inner"),
              ([4, 10], [6, 10], "inner"), ([3, 3, 5], [3, 2, 6], "all"), ([37], [41], "inner"), ([2, 1000], [2, 1000], "inner")]
    my_len, his_len, what = shapes[case % 8]
    maps = []
    for d, (ml, hl) in enumerate(zip(my_len, his_len)):
        remap = ml != hl or what == "all" or (what == "outer" and d == 0) or (what == "inner" and d == len(my_len) - 1)
        if remap:
            m_ = rng.permutation(max(ml, hl))[:hl]
            m_ = np.where(m_ < ml, m_, -1)
        else:
            m_ = np.arange(hl)
        maps.append(m_.astype(np.int32))
    type_name = ["float32", "float64", "int32", "uint32"][(case // 8 + case) % 4]
    my_default = float("nan") if case % 3 == 0 else 0.0
    his_default = float("nan") if case % 2 else 0.0
    n_my, n_his = int(np.prod(my_len)), int(np.prod(his_len))
    mine = np.where(rng.random(n_my) < 0.5, my_default, rng.integers(1, 50, size=n_my).astype(np.float64))
    his = np.where(rng.random(n_his) < 0.4, his_default, rng.integers(50, 99, size=n_his).astype(np.float64))

    def both(n, dflt, dense):
        o = OracleStore(n, type_name, dflt)
        typed = to_typed(dense, type_name).astype(np.float64)
        if type_name in ("int32", "uint32") and dflt != dflt:
            typed = np.where(np.isnan(dense), np.nan, typed)
        o.set_data(typed)
        g = pkg.HipStore(n, type_name, dflt)
        g.set_data_f64(dense)
        return o, g

    om, gm = both(n_my, my_default, mine)
    oh, gh = both(n_his, his_default, his)
    om.load(oh, my_len, his_len, maps)
    gm.load(gh, my_len, his_len, maps)
    ev, es = expected_typed(om)
    assert np.array_equal(gm.get_status(), es)
    assert same_typed(gm.get_data(), ev)


@pytest.mark.parametrize("type_name", ["float32", "float64", "int32", "uint32"])
@pytest.mark.parametrize("my_len,tag", [
    ([700, 10], "many rows per tile, tiles that end inside the cube"),
    ([3, 5, 4095], "one odd row per tile: rows that start at any cell offset"),
    ([5, 4096], "a row is exactly a tile (4-byte cells)"),
    ([2, 3, 1365], "three rows per tile, odd length"),
    ([1, 2], "one row of two cells"),
    ([9000, 3], "1 365 rows per tile"),
])
def test_load_permuted_rows(type_name, my_len, tag, monkeypatch):
    """load() from a store whose LAST dimension lists the same items in another order, nothing else remapped
    (load_permute_rows_kernel: rows rearranged through LDS, 16-byte loads and stores), every default pairing and both
    masks, against the oracle — and the same answer from the scatter form it replaces."""
    rng = np.random.default_rng(len(my_len) * 1000 + my_len[-1])
    cap = 16384 // np.dtype(type_name).itemsize
    maps = [np.arange(l, dtype=np.int32) for l in my_len[:-1]] + [rng.permutation(my_len[-1]).astype(np.int32)]
    n = int(np.prod(my_len))
    for my_default, his_default in [(0.0, 0.0), (float("nan"), 0.0), (0.0, float("nan")), (float("nan"), float("nan"))]:
        monkeypatch.delenv("OLAP_LOAD_NO_PERMUTE", raising=False)
        plan = pkg.Plan.load(type_name, my_default, his_default, my_len, my_len, maps)
        assert plan.kernel_name == ("load_permute_rows_kernel" if my_len[-1] <= cap else "load_scatter (16-byte runs of the other store)"), tag
        mine = np.where(rng.random(n) < 0.5, my_default, rng.integers(1, 50, size=n).astype(np.float64))
        his = np.where(rng.random(n) < 0.4, his_default, rng.integers(50, 99, size=n).astype(np.float64))
        results = []
        for no_permute in (False, True):
            if no_permute:
                monkeypatch.setenv("OLAP_LOAD_NO_PERMUTE", "1")
            else:
                monkeypatch.delenv("OLAP_LOAD_NO_PERMUTE", raising=False)

            def both(dflt, dense):
                o = OracleStore(n, type_name, dflt)
                typed = to_typed(dense, type_name).astype(np.float64)
                if type_name in ("int32", "uint32") and dflt != dflt:
                    typed = np.where(np.isnan(dense), np.nan, typed)
                o.set_data(typed)
                g = pkg.HipStore(n, type_name, dflt)
                g.set_data_f64(dense)
                return o, g

            om, gm = both(my_default, mine)
            oh, gh = both(his_default, his)
            om.load(oh, my_len, my_len, maps)
            gm.load(gh, my_len, my_len, maps)
            ev, es = expected_typed(om)
            assert np.array_equal(gm.get_status(), es), (tag, my_default, his_default, no_permute)
            assert same_typed(gm.get_data(), ev), (tag, my_default, his_default, no_permute)
            results.append(gm.get_data())
        assert same_typed(results[0], results[1])


@pytest.mark.parametrize("seed", range(40))
def test_drilldown_with_distributions_randomized(seed):
    """drillDown with per-cell weights (in-memory.js:389-401) on one or two refined dimensions."""
    rng = np.random.default_rng(8000 + seed)
    old_len = [int(rng.choice([1, 2, 3, 6])), int(rng.choice([1, 2, 5])), int(rng.choice([1, 4, 33]))]
    child_maps, new_len = [], []
    for d, l in enumerate(old_len):
        if d == 2 or rng.random() < 0.4:
            c = np.arange(l)
        else:
            c = np.repeat(np.arange(l), rng.integers(1, 5, size=l))
            if rng.random() < 0.5:
                c = rng.permutation(c)
                first = {}
                c = np.array([first.setdefault(int(x), len(first)) for x in c])  # parents numbered by first child
                if len(first) != l:
                    c = np.repeat(np.arange(l), 2)
        child_maps.append(c.astype(np.uint32))
        new_len.append(len(c))
    type_name = ["float32", "int32", "float64", "uint32"][seed % 4]
    default = float("nan") if seed % 2 else 0.0
    n_old, n_new = int(np.prod(old_len)), int(np.prod(new_len))
    vals = rng.integers(1, 200, size=n_old).astype(np.float64)
    dense = np.where(rng.random(n_old) < 0.25, default, vals)
    weights = rng.integers(1, 9, size=n_new).astype(np.float64) / 8.0
    o = OracleStore(n_old, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    g = pkg.HipStore(n_old, type_name, default)
    g.set_data_f64(dense)
    try:
        expected = o.drill_down(old_len, new_len, child_maps, "sum", weights)
    except ValueError as err:
        # the weight index of in-memory.js:392-396 assumes one added trailing dimension; with uneven
        # fan-outs it runs off the array and the reference throws — so must the device path, same index
        with pytest.raises(pkg.OlapError, match=str(err)):
            g.drill_down(old_len, new_len, child_maps, "sum", weights)
        return
    ev, es = expected_typed(expected)
    out = g.drill_down(old_len, new_len, child_maps, "sum", weights)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("method", ["sum", "average"])
@pytest.mark.parametrize("type_name", ["float32", "int32"])
@pytest.mark.parametrize("K", [257, 1000])
def test_tile_regime_long_last_dimension(K, type_name, method):
    """Rolling up a long last dimension with more than 131 072 rows: the tile kernel's 16-lanes-per-row
    path (float64, re-associated like the reduce regime; the inputs are half-integers, so exact here)."""
    rng = np.random.default_rng(41)
    lens = [131500, K]
    n = lens[0] * K
    vals = rng.integers(-8, 9, size=n).astype(np.float64) * (0.5 if type_name == "float32" else 1.0)
    dense = np.where(rng.random(n) < 0.3, 0.0, vals)
    maps = [np.arange(lens[0], dtype=np.uint32), np.zeros(K, np.uint32)]
    plan = pkg.Plan.drillup(type_name, 0.0, method, lens, [lens[0], 1], maps)
    assert plan.kernel_name == "drillup_tile_kernel", plan.kernel_name
    g = pkg.HipStore(n, type_name, 0.0)
    g.set_data_f64(dense)
    out = g.drill_up(lens, [lens[0], 1], maps, method)
    typed = to_typed(dense, type_name).astype(np.float64).reshape(lens)
    total = typed.sum(axis=1)
    count = (typed != 0).sum(axis=1)
    expect = total if method == "sum" else np.where(count > 0, total / np.maximum(count, 1), 0.0)
    ev = to_typed(expect, type_name)
    assert same_typed(out.get_data(), ev)
    assert np.array_equal(out.get_status() == 2, ev != 0)


@pytest.mark.parametrize("method", ["sum", "average", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("float64", 0.0), ("uint32", float("nan")), ("int32", 0.0)])
@pytest.mark.parametrize("outer,K,inner,groups", [
    (1003, 1000, 1, "mod10"),      # 4 rows x 10 runs of 100 per tile (runs padded onto distinct banks), last tile partial
    (1001, 1000, 1, "mod100"),     # 100 runs of 10: the padding costs a row of the tile (3 rows instead of 4)
    (501, 100, 10, "mod10"),       # inner = 10: a member is ten cells
    (77, 4096, 1, "random"),       # a row fills the whole tile (float32; two tiles' worth for float64 -> another regime)
    (1999, 37, 3, "random"),       # odd everything: rows of 111 cells, ragged runs
    (260, 613, 5, "ragged"),       # runs of very different lengths
])
def test_tile_regime_permuted_groups(outer, K, inner, groups, type_name, default, method):
    """Interleaved groups in the row-tile regime (drillup_tile_kernel MODE 3): the cells of a row are permuted into
    group order on their way into LDS, runs padded onto distinct banks when that fits.  Bit-exact against the oracle."""
    rng = np.random.default_rng(K * 31 + inner)
    if groups == "mod10":
        amap = (np.arange(K) % 10).astype(np.uint32)
    elif groups == "mod100":
        amap = (np.arange(K) % 100).astype(np.uint32)
    elif groups == "ragged":
        raw = np.minimum(rng.geometric(0.15, size=K) - 1, 20)
        first = {}
        amap = np.array([first.setdefault(int(g), len(first)) for g in raw], dtype=np.uint32)
    else:
        raw = rng.integers(0, max(2, K // 7), size=K)
        first = {}
        amap = np.array([first.setdefault(int(g), len(first)) for g in raw], dtype=np.uint32)
    lens = [outer, K, inner]
    new = [outer, int(amap.max()) + 1, inner]
    n = outer * K * inner
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) if type_name != "uint32" else np.ones(n)
        vals = vals * np.where(rng.random(n) < 4.0 / K, 2.0, 1.0)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=n).astype(np.float64)
        if type_name.startswith("float"):
            vals = vals * 0.5
    dense = np.where(rng.random(n) < 0.3, default, vals)
    maps = [np.arange(outer, dtype=np.uint32), amap, np.arange(inner, dtype=np.uint32)]
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    if K * inner * (8 if type_name == "float64" else 4) <= 16384:
        assert plan.kernel_name == "drillup_tile_kernel", plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es), plan.kernel_name
    assert same_typed(out.get_data(), ev), plan.kernel_name


@pytest.mark.parametrize("lens,axis", [([7, 9, 513], 1), ([3, 30, 1001], 1), ([5, 4, 2049], 1), ([12, 3, 171], 0), ([2, 40, 515], 1),
                                       ([300, 7, 65], 1), ([3, 5, 7, 33], 1), ([1, 6, 1023], 1)])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("uint32", float("nan")), ("float64", 0.0), ("int32", 0.0)])
@pytest.mark.parametrize("messy", [False, True])
def test_dice_row_copies_odd_extents(lens, axis, type_name, default, messy):
    """dice of one dimension of a cube with odd extents, every new item naming a distinct old item
    (subset, reordered), with and without an unknown item and an old item named twice: dice_direct_kernel
    instead of the 4-byte gather."""
    rng = np.random.default_rng(53)
    n = int(np.prod(lens))
    vals = rng.integers(1, 500, size=n).astype(np.float64)
    dense = np.where(rng.random(n) < 0.3, default, vals)
    keep = rng.permutation(lens[axis])[: max(1, lens[axis] * 2 // 3)]
    if lens[0] % 2:
        keep = np.sort(keep)
    if messy:  # an unknown item and an old item named twice (only its last mention receives the cells)
        keep = np.concatenate([keep[:1], [-1], keep, keep[:1]])
    sel = [np.arange(l, dtype=np.int32) for l in lens]
    sel[axis] = keep.astype(np.int32)
    new_len = [len(x) for x in sel]
    plan = pkg.Plan.dice(type_name, default, lens, new_len, sel)
    inner = int(np.prod(lens[axis + 1:]))
    isz = np.dtype(type_name).itemsize
    odd = inner % (16 // isz) != 0
    if odd and inner >= 16 // isz:
        expected = "dice_direct_kernel"  # aligned destination groups, one cell-aligned 16-byte load each
    else:
        expected = "gather(dice)"
    assert plan.kernel_name == expected, plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.dice(lens, new_len, sel))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.dice(lens, new_len, sel)
    assert np.array_equal(out.get_status(), es)
    assert same_typed(out.get_data(), ev)


REORDER_CASES = [
    # lens, perm, kernel of the masked / 8-byte forms, whether 4-byte cells without a mask take the two-axis transpose
    ([10] * 6, [5, 4, 3, 2, 1, 0], "reorder_brick4_kernel", True),    # runs of 100 cells on both sides
    ([12, 7, 20], [2, 1, 0], "reorder_brick4_kernel", False),         # one brick = the whole cube (Y axis of 12 cells: too short)
    ([64, 48], [1, 0], "reorder_brick4_kernel", True),                # partial dimension on the read side
    ([1000, 1000], [1, 0], "reorder_brick4_kernel", True),            # 100 x 100 bricks
    ([50, 100, 1000], [2, 0, 1], None, True),
    ([37, 53], [1, 0], "reorder_brick_kernel", True),                 # odd extents: ragged scalar bricks / 4-byte tiles
    ([6, 1, 5, 4], [3, 0, 2, 1], None, False),
    ([3, 250, 9, 30], [1, 3, 0, 2], None, None),
    ([20, 30, 40], [0, 2, 1], None, True),                            # leading dimension untouched
    ([20, 30, 40], [1, 0, 2], "gather(reorder)", False),              # fastest dimension untouched: 16 B gather
    ([130, 3, 131], [2, 1, 0], None, True),                           # tiles that end inside a dimension, edge tiles on both axes
    ([5, 300, 7, 260], [3, 1, 2, 0], None, True),                     # batch dimensions on both sides of the axes
]


@pytest.mark.parametrize("lens,perm,kernel,xy", REORDER_CASES)
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", float("nan")),
                                               ("float64", 0.0)])
def test_reorder_forms(lens, perm, kernel, xy, type_name, default):
    """reorder (in-memory.js:178-211) against numpy's transpose: the two-axis LDS transpose (4- and 8-byte cells without
    a mask), the 16-byte brick form, the scalar (ragged) brick form and the gather form, with and without the mask."""
    rng = np.random.default_rng(29)
    n = int(np.prod(lens))
    vals = rng.integers(1, 1000, size=n).astype(np.float64)
    unset = rng.random(n) < 0.3
    dense = np.where(unset, default, vals)
    plan = pkg.Plan.reorder(type_name, default, lens, perm)
    four_bytes = type_name != "float64"
    if xy is not None:
        assert (plan.kernel_name == "transpose_xy_kernel") == xy, plan.kernel_name
    if kernel is not None and not xy and (four_bytes or "brick4" not in kernel):
        assert plan.kernel_name == kernel, plan.kernel_name
    if not four_bytes:
        assert plan.kernel_name != "reorder_brick4_kernel"
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.reorder(lens, perm)
    moved = lambda a: np.ascontiguousarray(a.reshape(lens).transpose(perm)).ravel()  # noqa: E731
    assert np.array_equal(out.get_status(), moved(g.get_status()))
    assert same_typed(out.get_data(), moved(g.get_data()))
    assert np.array_equal(out.get_status() == 2, moved(~unset))
    # raw pointers: the mask generated on the way out (no mask given on the way in), where the values can tell
    if not (default != default and type_name in ("int32", "uint32")):
        o2 = pkg.HipStore(n, type_name, default)
        plan.run(g.values_ptr, None, o2.values_ptr, o2.status_ptr)
        pkg.capi.check(pkg.lib().olap_device_synchronize())
        assert same_typed(o2.get_data(), moved(g.get_data())) and np.array_equal(o2.get_status(), moved(g.get_status()))


@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("uint32", float("nan")), ("float64", 0.0), ("int32", 0.0)])
def test_sparse_form_round_trip(type_name, default):
    """Device-side stream compaction (the reference's serialised layout, in-memory.js:94-100) and back."""
    rng = np.random.default_rng(3)
    for n in (0, 1, 63, 64, 65, 1000, 300_001):
        vals = rng.integers(1, 100, size=n).astype(np.float64)
        unset = rng.random(n) < 0.6
        dense = np.where(unset, default, vals)
        s = pkg.HipStore(n, type_name, default)
        s.set_data_f64(dense)
        idx, v = s.to_sparse()
        keep = np.nonzero(~unset)[0]
        assert np.array_equal(idx, keep.astype(np.uint32))
        assert np.array_equal(v.astype(np.float64), vals[keep])
        back = pkg.HipStore.from_sparse(n, type_name, default, idx, v)
        assert same_typed(back.get_data(), s.get_data()) and np.array_equal(back.get_status(), s.get_status())


def _mulberry_cell_values(cells, seed=20240807):
    """fround(0.5 + u(2*cell + 1)) for arbitrary (64-bit) cell indices — the device generator in closed form."""
    cells = np.asarray(cells, dtype=np.uint64)
    a = ((np.uint64(seed) + (np.uint64(2) * cells + np.uint64(1)) * np.uint64(0x6D2B79F5)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    with np.errstate(over="ignore"):
        t = (a ^ (a >> np.uint32(15))) * (np.uint32(1) | a)
        t = (t + ((t ^ (t >> np.uint32(7))) * (np.uint32(61) | t))) ^ t
        r = t ^ (t >> np.uint32(14))
    return (0.5 + r.astype(np.float64) / 4294967296.0).astype(np.float32)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_five_billion_cells_64bit_indexing(axis):
    """A 5x10^9-cell measure (20 GB, flat indices beyond 2^32): drillUp of each axis, spot-checked
    against the closed form of the on-device generator (no 20 GB download)."""
    shape = [5000, 1000, 1000]
    n = int(np.prod(shape))
    L = pkg.lib()
    s = pkg.HipStore(n, "float32", 0.0)
    pkg.capi.check(L.olap_fill_seeded(s.values_ptr, None, n, 0, 2, 20240807, 1.0, None))
    new_len = list(shape)
    new_len[axis] = 1
    maps = [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(shape)]
    out = s.drill_up(shape, new_len, maps, "sum")
    got = out.get_data()
    assert got.size == n // shape[axis]
    rng = np.random.default_rng(axis)
    strides = [shape[1] * shape[2], shape[2], 1]
    out_shape = list(new_len)
    for flat in list(rng.integers(0, got.size, size=24)) + [0, got.size - 1]:
        idx = list(np.unravel_index(int(flat), out_shape))
        base = sum(int(idx[d]) * strides[d] for d in range(3) if d != axis)
        cells = np.uint64(base) + np.arange(shape[axis], dtype=np.uint64) * np.uint64(strides[axis])
        expect = np.float32(np.sum(_mulberry_cell_values(cells).astype(np.float64)))  # ascending k, float64
        ref = 0.0
        for v in _mulberry_cell_values(cells).astype(np.float64):
            ref += v
        assert got[int(flat)] == np.float32(ref), (axis, flat, got[int(flat)], ref, expect)
    del s, out


def test_empty_and_degenerate_shapes():
    """Zero-length dimensions, zero-dimensional cubes and empty selections (test/cube-filtering.js:98-100)."""
    s = pkg.HipStore(6, "float32", 0.0)
    s.set_data(np.array([1, 2, 4, 8, 16, 32], np.float32))
    empty = s.dice([3, 2], [0, 2], [np.zeros(0, np.int32), np.arange(2, dtype=np.int32)])
    assert empty.size == 0 and empty.get_data().size == 0 and empty.count_set() == 0 and empty.total == 0
    # rolling up an empty dimension gives its one 'all' item, unset
    up = empty.drill_up([0, 2], [1, 2], [np.zeros(0, np.uint32), np.arange(2, dtype=np.uint32)], "sum")
    assert up.size == 2 and np.array_equal(up.get_data(), [0, 0]) and up.count_set() == 0
    scalar = pkg.HipStore(1, "float32", 0.0)
    scalar.set_data(np.array([32], np.float32))
    same = scalar.drill_up([], [], [], "average")
    assert np.array_equal(same.get_data(), [32])
    assert np.array_equal(scalar.reorder([], []).get_data(), [32])
    zero = pkg.HipStore(0, "int32", float("nan"))
    assert zero.get_data().size == 0 and zero.keys().size == 0
    idx, vals = zero.to_sparse()
    assert idx.size == 0 and vals.size == 0


def test_plans_replay_inside_a_hip_graph():
    """olap_plan_run is pure kernel launches on the given stream, so a chain of plans can be
    captured once into a hipGraph (through torch.cuda.CUDAGraph) and replayed: collapse() of a
    6-dimensional cube = six drillUps, one graph launch."""
    import torch

    shape = [10] * 6
    n = 10 ** 6
    vals, _ = config_cube(n, 5, 1.0)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(vals).to(dev)
    plans, bufs, lens = [], [src], list(shape)
    for axis in range(6):
        new = list(lens)
        new[axis] = 1
        maps = [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
        plans.append(pkg.Plan.drillup("float32", 0.0, "sum", lens, new, maps))
        bufs.append(torch.empty(int(np.prod(new)), dtype=torch.float32, device=dev))
        lens = new
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for p, a, b in zip(plans, bufs[:-1], bufs[1:]):  # warm-up outside capture
            p.run(a.data_ptr(), None, b.data_ptr(), None, stream.cuda_stream)
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for p, a, b in zip(plans, bufs[:-1], bufs[1:]):
                p.run(a.data_ptr(), None, b.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    eager = bufs[-1].clone()
    src.mul_(2.0)  # new input, same graph
    graph.replay()
    torch.cuda.synchronize()
    assert torch.allclose(bufs[-1], eager * 2, rtol=1e-6)
    assert abs(float(eager.item()) - float(vals.astype(np.float64).sum())) <= 1e-5 * float(vals.sum())


@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)])
@pytest.mark.parametrize("method", ["sum", "average"])
def test_drilldown_row_form(type_name, default, method):
    """One refined axis with wide rows (drilldown_rows_kernel): contiguous and interleaved child lists,
    integer remainder spreading by child ordinal, unset / zero / NaN parents skipped."""
    rng = np.random.default_rng(21)
    # inner 520 / 2600: rows off the 128-byte grid (float cells take drilldown_rows_lines_kernel, one and
    # several windows per row); 1024: line-aligned rows; 19 children: more than one segment per parent
    cases = [(np.repeat(np.arange(4), [3, 1, 5, 3]), 520), (np.arange(12) % 4, 520), (np.arange(12) % 4, 2600),
             (np.repeat(np.arange(4), [3, 1, 5, 3]), 1024), (np.repeat(np.arange(2), [19, 2]), 2056)]
    for child_map, inner in cases:
        G, K = int(child_map.max()) + 1, len(child_map)
        old_len, new_len = [3, G, inner], [3, K, inner]
        n_old = int(np.prod(old_len))
        vals = rng.integers(-40, 90, size=n_old).astype(np.float64)
        if type_name == "uint32":
            vals = np.abs(vals)
        if type_name.startswith("float"):
            vals = vals * 0.5
        dense = np.where(rng.random(n_old) < 0.25, default, vals)
        maps = [np.arange(3, dtype=np.uint32), child_map.astype(np.uint32), np.arange(inner, dtype=np.uint32)]
        plan = pkg.Plan.drilldown(type_name, default, method, old_len, new_len, maps)
        lines = (type_name.startswith("float") or method != "sum") and (inner * np.dtype(type_name).itemsize) % 128 != 0
        assert plan.kernel_name == ("drilldown_rows_lines_kernel" if lines else "drilldown_rows_kernel")
        o = OracleStore(n_old, type_name, default)
        typed = to_typed(dense, type_name).astype(np.float64)
        if type_name in ("int32", "uint32") and default != default:
            typed = np.where(np.isnan(dense), np.nan, typed)
        o.set_data(typed)
        ev, es = expected_typed(o.drill_down(old_len, new_len, maps, method))
        g = pkg.HipStore(n_old, type_name, default)
        g.set_data_f64(dense)
        out = g.drill_down(old_len, new_len, maps, method)
        assert np.array_equal(out.get_status(), es)
        assert same_typed(out.get_data(), ev)


@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("uint32", float("nan")), ("float64", 0.0), ("int32", 0.0)])
@pytest.mark.parametrize("lens,axis,kind,method", [
    ([6, 40, 1100], 0, "all", "sum"),            # row regime, 16-byte lanes
    ([9, 30, 1027], 1, "interleaved", "last"),   # row regime, ragged slots
    ([40, 30, 200], 1, "contiguous", "average"),  # flat regime
    ([300, 100, 3], 1, "interleaved", "sum"),     # row tile, permuted groups
    ([300, 100, 3], 1, "all", "highest"),         # row tile, one group
    ([3, 9000, 5], 1, "thirty", "sum"),           # group tiles
    ([2, 5000, 2], 1, "all", "sum"),              # few outputs, long groups: the reduce regime runs pair by pair
    ([5, 4, 3], 2, "all", "product"),
])
@pytest.mark.parametrize("n", [1, 3, 8, 11])
def test_drillup_batch_matches_single(lens, axis, kind, method, type_name, default, n):
    """olap_store_drillup_batch / olap_plan_run_batch: n measures in one call (one launch per 8 where the plan allows
    it) give exactly the stores that n single calls give; the first is also checked against the oracle."""
    rng = np.random.default_rng(sum(lens) * 7 + n)
    K = lens[axis]
    amap = {"all": np.zeros(K), "interleaved": np.arange(K) % 7, "contiguous": np.arange(K) // 4, "thirty": np.arange(K) // 30}[kind].astype(np.uint32)
    new = list(lens)
    new[axis] = int(amap.max()) + 1
    maps = [amap if d == axis else np.arange(l, dtype=np.uint32) for d, l in enumerate(lens)]
    cells = int(np.prod(lens))
    stores, denses = [], []
    for _ in range(n):
        if method == "product":
            vals = np.where(rng.random(cells) < 0.5, 1.0, 2.0)
        else:
            vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=cells).astype(np.float64)
        dense = np.where(rng.random(cells) < 0.3, default, vals)
        g = pkg.HipStore(cells, type_name, default)
        g.set_data_f64(dense)
        stores.append(g)
        denses.append(dense)
    batch = pkg.HipStore.drill_up_batch(stores, lens, new, maps, method)
    assert len(batch) == n
    for g, b in zip(stores, batch):
        single = g.drill_up(lens, new, maps, method)
        assert np.array_equal(b.get_status(), single.get_status())
        assert same_typed(b.get_data(), single.get_data())
    o = OracleStore(cells, type_name, default)
    typed = to_typed(denses[0], type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(denses[0]), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    if not (kind == "all" and K >= 256 and method in ("sum", "average", "product")):  # (re-associated regimes: checked above against the single call)
        assert np.array_equal(batch[0].get_status(), es)
        assert same_typed(batch[0].get_data(), ev)


def test_drillup_batch_falls_back_and_validates():
    """Stores that differ in cell type, or track their insertion order, are rolled up one by one behind the same call;
    raw-pointer batches with masks on some pairs only run pair by pair; a NULL in the list is refused."""
    lens, new = [4, 6, 5], [1, 6, 5]
    maps = [np.zeros(4, np.uint32), np.arange(6, dtype=np.uint32), np.arange(5, dtype=np.uint32)]
    rng = np.random.default_rng(5)
    a, b, c = pkg.HipStore(120, "float32", 0.0), pkg.HipStore(120, "int32", 0.0), pkg.HipStore(120, "float32", 0.0)
    for s in (a, b, c):
        s.set_data_f64(rng.integers(-3, 4, size=120).astype(np.float64))
    c.track_order(True)
    c.set_value(7, 0.0)
    c.set_value(7, 9.0)  # leaves the flat index order: the tracked path
    outs = pkg.HipStore.drill_up_batch([a, b, c], lens, new, maps, "first")
    for s, o in zip((a, b, c), outs):
        single = s.drill_up(lens, new, maps, "first")
        assert np.array_equal(o.get_data(), single.get_data()) and np.array_equal(o.get_status(), single.get_status())
    assert pkg.HipStore.drill_up_batch([], lens, new, maps, "sum") == []
    L = pkg.lib()
    hs = (C.c_void_p * 2)(a._h, None)
    outs2 = (C.c_void_p * 2)()
    ol, nl = np.asarray(lens, np.uint32), np.asarray(new, np.uint32)
    keep = [np.ascontiguousarray(m) for m in maps]
    arr = (capi._pu32 * 3)(*[m.ctypes.data_as(capi._pu32) for m in keep])
    rc = L.olap_store_drillup_batch(2, hs, outs2, 3, ol.ctypes.data_as(capi._pu32), nl.ctypes.data_as(capi._pu32), arr, 0)
    assert rc != 0 and b"NULL" in L.olap_last_error() and outs2[0] is None


@pytest.mark.parametrize("lens,gmap_mod", [([7, 12, 20], 5), ([3, 40, 1028], 7), ([50, 64, 6], 4)])
def test_plan_run_batch_on_raw_pointers(lens, gmap_mod):
    """olap_plan_run_batch on raw device pointers: aligned buffers, buffers 4 bytes off a 16-byte boundary (the whole
    batch then moves cell by cell), masks on every pair, and masks on SOME pairs only (pair by pair behind the call) —
    the same cells as single runs."""
    import torch

    n = int(np.prod(lens))
    K = lens[1]
    maps = [np.arange(lens[0], dtype=np.uint32), (np.arange(K) % gmap_mod).astype(np.uint32), np.arange(lens[2], dtype=np.uint32)]
    new = [lens[0], gmap_mod, lens[2]]
    nm = 5
    rng = np.random.default_rng(n)
    for method in ("sum", "last"):
        plan = pkg.Plan.drillup("float32", 0.0, method, lens, new, maps)
        n_out = plan.out_cells
        srcs = [pkg.HipStore(n, "float32", 0.0) for _ in range(nm)]
        for s in srcs:
            s.set_data_f64(np.where(rng.random(n) < 0.3, 0.0, rng.integers(-8, 9, size=n) * 0.5))
        want = []
        for s in srcs:
            o = pkg.HipStore(n_out, "float32", 0.0)
            plan.run(s.values_ptr, None, o.values_ptr, o.status_ptr)
            want.append((o.get_data(), o.get_status()))
        for off in (0, 1):  # cells off a 16-byte boundary
            ins = [torch.empty(n + 4, dtype=torch.float32, device="cuda") for _ in range(nm)]
            outs = [torch.empty(n_out + 4, dtype=torch.float32, device="cuda") for _ in range(nm)]
            sts = [torch.empty(n_out + 4, dtype=torch.int32, device="cuda") for _ in range(nm)]
            for t, s in zip(ins, srcs):
                t[off:off + n].copy_(torch.from_numpy(s.get_data()).cuda())
            torch.cuda.synchronize()
            plan.run_batch([t.data_ptr() + 4 * off for t in ins], None, [t.data_ptr() + 4 * off for t in outs], [t.data_ptr() + 4 * off for t in sts])
            torch.cuda.synchronize()
            for i in range(nm):
                assert same_typed(outs[i][off:off + n_out].cpu().numpy(), want[i][0]), (method, off, i)
                assert np.array_equal(sts[i][off:off + n_out].cpu().numpy(), want[i][1]), (method, off, i)
        # masks on every pair / on some pairs only
        outs = [pkg.HipStore(n_out, "float32", 0.0) for _ in range(nm)]
        plan.run_batch([s.values_ptr for s in srcs], [s.status_ptr for s in srcs], [o.values_ptr for o in outs], [o.status_ptr for o in outs])
        pkg.capi.check(pkg.lib().olap_device_synchronize())
        for i in range(nm):
            assert same_typed(outs[i].get_data(), want[i][0]) and np.array_equal(outs[i].get_status(), want[i][1])
        outs = [pkg.HipStore(n_out, "float32", 0.0) for _ in range(nm)]
        plan.run_batch([s.values_ptr for s in srcs], [s.status_ptr if i % 2 else 0 for i, s in enumerate(srcs)], [o.values_ptr for o in outs],
                       [o.status_ptr for o in outs])
        pkg.capi.check(pkg.lib().olap_device_synchronize())
        for i in range(nm):
            assert same_typed(outs[i].get_data(), want[i][0]) and np.array_equal(outs[i].get_status(), want[i][1])


RULES = ["sum", "average", "highest", "lowest", "first", "last", "product"]


@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("uint32", float("nan")), ("float64", 0.0), ("int32", 0.0)])
@pytest.mark.parametrize("lens,axis,kind", [
    ([6, 40, 1100], 0, "all"),            # row regime, 16-byte lanes, wide rows: one mixed-rule launch
    ([9, 30, 4100], 1, "interleaved"),    # row regime, one row in flight, member list
    ([12, 31, 520], 1, "contiguous"),     # row regime, four rows in flight
    ([20, 100, 274], 1, "contiguous"),    # rows of an even number of cells: 8-byte lanes (config 5's city -> country)
    ([9, 30, 1027], 1, "interleaved"),    # ragged rows: rule by rule
    ([300, 100, 3], 1, "interleaved"),    # row tile, permuted groups: one mixed-rule launch (drillup_tile_mixed_kernel)
    ([200, 60, 7], 1, "contiguous"),      # row tile, contiguous groups
    ([500, 12, 9], 1, "all"),             # row tile, one group
    ([40, 30, 200], 1, "contiguous"),     # flat regime: rule by rule
    ([2, 5000, 2], 1, "all"),             # reduce regime: rule by rule
])
@pytest.mark.parametrize("n", [2, 4, 9])
def test_drillup_multi_rules_matches_single(lens, axis, kind, type_name, default, n):
    """olap_store_drillup_multi: n measures with a rule EACH in one call — one mixed-rule launch in the row and row-tile
    regimes (drillup_rows_mixed_kernel, drillup_tile_mixed_kernel), one launch per rule elsewhere — give exactly the
    stores n single calls give."""
    rng = np.random.default_rng(sum(lens) * 13 + n)
    K = lens[axis]
    amap = {"all": np.zeros(K), "interleaved": np.arange(K) % 7, "contiguous": np.arange(K) // 4}[kind].astype(np.uint32)
    new = list(lens)
    new[axis] = int(amap.max()) + 1
    maps = [amap if d == axis else np.arange(l, dtype=np.uint32) for d, l in enumerate(lens)]
    cells = int(np.prod(lens))
    rules = [RULES[i % 7] for i in rng.permutation(max(n, 7))[:n]]
    if n >= 4:
        rules[:4] = ["sum", "average", "first", "last"]  # config 5's
    stores = []
    for rule in rules:
        if rule == "product":
            vals = np.where(rng.random(cells) < 0.5, 1.0, 2.0)
        else:
            vals = rng.integers(0 if type_name == "uint32" else -8, 9, size=cells).astype(np.float64)
        g = pkg.HipStore(cells, type_name, default)
        g.set_data_f64(np.where(rng.random(cells) < 0.3, default, vals))
        stores.append(g)
    multi = pkg.HipStore.drill_up_multi(stores, rules, lens, new, maps)
    assert len(multi) == n
    for g, rule, m in zip(stores, rules, multi):
        single = g.drill_up(lens, new, maps, rule)
        assert np.array_equal(m.get_status(), single.get_status()), rule
        assert same_typed(m.get_data(), single.get_data()), rule


def test_drillup_multi_mixed_stores_and_plan_level():
    """Stores of different cell types in one olap_store_drillup_multi call (grouped behind it); olap_plan_run_batch_rules
    on raw pointers: one mixed-rule launch, and pair by pair when the buffers are off a 16-byte boundary."""
    import torch

    lens, new = [5, 20, 1200], [5, 4, 1200]
    maps = [np.arange(5, dtype=np.uint32), (np.arange(20) % 4).astype(np.uint32), np.arange(1200, dtype=np.uint32)]
    n = int(np.prod(lens))
    rng = np.random.default_rng(77)
    kinds = [("float32", 0.0, "sum"), ("int32", 0.0, "last"), ("float32", 0.0, "highest"), ("float64", 0.0, "average"), ("int32", 0.0, "sum"),
             ("float32", float("nan"), "first")]
    stores = []
    for t, d, _r in kinds:
        g = pkg.HipStore(n, t, d)
        g.set_data_f64(np.where(rng.random(n) < 0.3, d, rng.integers(-8, 9, size=n).astype(np.float64)))
        stores.append(g)
    outs = pkg.HipStore.drill_up_multi(stores, [k[2] for k in kinds], lens, new, maps)
    for g, k, o in zip(stores, kinds, outs):
        single = g.drill_up(lens, new, maps, k[2])
        assert same_typed(o.get_data(), single.get_data()) and np.array_equal(o.get_status(), single.get_status()), k
    with pytest.raises(pkg.OlapError) as ei:
        pkg.HipStore.drill_up_multi(stores[:2], ["sum", "median"], lens, new, maps)
    assert "Unsupported aggregation method" in str(ei.value)
    # plan level
    rules = ["sum", "average", "first", "last", "product"]
    srcs = [stores[0], stores[2]] + [pkg.HipStore(n, "float32", 0.0) for _ in range(3)]
    for s_ in srcs[2:]:
        s_.set_data_f64(np.where(rng.random(n) < 0.3, 0.0, np.where(rng.random(n) < 0.5, 1.0, 2.0)))
    plan = pkg.Plan.drillup("float32", 0.0, "lowest", lens, new, maps)  # (the planned rule is ignored)
    n_out = plan.out_cells
    want = [s_.drill_up(lens, new, maps, r) for s_, r in zip(srcs, rules)]
    for off in (0, 1):
        ins = [torch.empty(n + 4, dtype=torch.float32, device="cuda") for _ in srcs]
        outs_t = [torch.empty(n_out + 4, dtype=torch.float32, device="cuda") for _ in srcs]
        sts = [torch.empty(n_out + 4, dtype=torch.int32, device="cuda") for _ in srcs]
        for t, s_ in zip(ins, srcs):
            t[off:off + n].copy_(torch.from_numpy(s_.get_data()).cuda())
        torch.cuda.synchronize()
        plan.run_batch_rules(rules, [t.data_ptr() + 4 * off for t in ins], None, [t.data_ptr() + 4 * off for t in outs_t],
                             [t.data_ptr() + 4 * off for t in sts])
        torch.cuda.synchronize()
        for i in range(len(srcs)):
            assert same_typed(outs_t[i][off:off + n_out].cpu().numpy(), want[i].get_data()), (off, rules[i])
            assert np.array_equal(sts[i][off:off + n_out].cpu().numpy(), want[i].get_status()), (off, rules[i])


@pytest.mark.parametrize("method", ["sum", "average", "highest", "first", "last", "product"])
@pytest.mark.parametrize("type_name,default", [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)])
@pytest.mark.parametrize("lens,axis,kind", [([3000, 132], 0, "all"), ([2, 1500, 256], 1, "all"), ([2600, 200], 0, "halves"), ([3, 900, 516], 1, "interleaved"),
                                            ([2100, 250], 0, "all"), ([1200, 1000], 0, "all"), ([1300, 1020], 0, "all"), ([1100, 2052], 0, "all"), ([2001, 251], 0, "all"),
                                            ([1500, 514], 0, "all")])
def test_split_regime_wide_rows(lens, axis, kind, type_name, default, method):
    """Few output cells, long groups, rows wider than 128 cells: segments of every group are streamed by lanes that own
    four adjacent output cells (drillup_split4_kernel: 4-byte cells, rows of a multiple of 4 cells) or one
    (drillup_split_kernel), then merged.  Contiguous groups take incremental addressing, interleaved ones the member list."""
    rng = np.random.default_rng(sum(lens) + len(method))
    K = lens[axis]
    amap = {"all": np.zeros(K), "halves": (np.arange(K) >= K // 3).astype(int), "interleaved": np.arange(K) % 3}[kind].astype(np.uint32)
    new = list(lens)
    new[axis] = int(amap.max()) + 1
    maps = [amap if d == axis else np.arange(l, dtype=np.uint32) for d, l in enumerate(lens)]
    n = int(np.prod(lens))
    if method == "product":
        vals = np.where(rng.random(n) < 0.5, 1.0, -1.0) * np.where(rng.random(n) < 0.004, 2.0, 1.0)
        if type_name == "uint32":
            vals = np.abs(vals)
    else:
        vals = rng.integers(0 if type_name == "uint32" else -40, 41, size=n).astype(np.float64)
        if type_name.startswith("float"):
            vals = vals * 0.25
    dense = np.where(rng.random(n) < 0.3, default, vals)
    plan = pkg.Plan.drillup(type_name, default, method, lens, new, maps)
    # one contiguous '-> all' group whose smallest whole-16-byte run of rows fits a workgroup's lanes (1 024 four-byte
    # cells, 512 eight-byte ones): the cooperative 16-byte form streams whole segments (drillup_reduce4_kernel);
    # everything else here takes the lane-per-cell split forms or the row kernel over segments
    inner = int(np.prod(lens[axis + 1:]))
    item = 8 if type_name == "float64" else 4
    per16 = 16 // item
    rows_min = next(r for r in (1, 2, 4) if (r * inner) % per16 == 0)
    coop = kind == "all" and rows_min * inner <= 256 * per16 and (K * inner) % per16 == 0
    vec = next(v for v in (16 // item, 2, 1) if inner % v == 0 and v <= 16 // item)
    if coop:
        assert "reduce4" in plan.kernel_name, plan.kernel_name
    elif method != "product" and inner * item >= 2048 and inner // vec >= 128:
        # rows of 2 KB and more: the row kernel over segments of the groups + a fold (SegmentedRows)
        assert "segments" in plan.kernel_name, plan.kernel_name
    else:
        assert "split" in plan.kernel_name, plan.kernel_name
        if type_name != "float64" and lens[-1] % 4 == 0:
            assert "split4" in plan.kernel_name, plan.kernel_name
    o = OracleStore(n, type_name, default)
    typed = to_typed(dense, type_name).astype(np.float64)
    if type_name in ("int32", "uint32") and default != default:
        typed = np.where(np.isnan(dense), np.nan, typed)
    o.set_data(typed)
    ev, es = expected_typed(o.drill_up(lens, new, maps, method))
    g = pkg.HipStore(n, type_name, default)
    g.set_data_f64(dense)
    out = g.drill_up(lens, new, maps, method)
    assert np.array_equal(out.get_status(), es)
    gv = out.get_data()
    if type_name == "float64" and method in ("sum", "average", "product"):
        assert np.allclose(gv, ev, rtol=1e-12, atol=0, equal_nan=True)
    else:
        assert same_typed(gv, ev)
