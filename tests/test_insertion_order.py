"""Tracked stores (olap_store_track_order) reproduce the reference Map's INSERTION order: `first` / `last`, keys() and
the sparse (serialized) form follow it through setValue, data=, fill, drillUp, dice, reorder, drillDown and load.
Expectations: the CPU oracle, which keeps the Map's insertion log (oracle/olap_oracle.c) and is pinned to the
reference's own outputs — including the two golden cases first_insertion_order / last_insertion_order."""
import numpy as np
import pytest

from conftest import load_package
from golden_util import dec_store, expected_typed, load_cases
from oracle.oracle import OracleStore

pytestmark = pytest.mark.gpu

pkg = load_package()
KAT = {c["name"]: c for c in load_cases("store_kat.json")}
ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731


def pair(n, type_name="float32", default=0.0):
    return OracleStore(n, type_name, default), pkg.HipStore(n, type_name, default).track_order()


def same(o, g, what=""):
    """values, mask AND key order"""
    ev, es = expected_typed(o)
    assert np.array_equal(g.get_status(), es), what
    assert np.array_equal(g.get_data().astype(np.float64), ev.astype(np.float64), equal_nan=True), what
    keys = o.entries()[0]
    keys = keys[es[keys.astype(np.int64)] == 2]  # (a typed store drops keys whose value rounds to the default)
    assert g.keys().tolist() == keys.tolist(), (what, g.keys().tolist(), keys.tolist())
    idx, vals = g.to_sparse()
    assert idx.tolist() == keys.tolist() and np.array_equal(vals.astype(np.float64), ev[keys.astype(np.int64)].astype(np.float64), equal_nan=True), what


@pytest.mark.parametrize("name", ["first_insertion_order", "last_insertion_order"])
def test_golden_insertion_order_cases(name):
    """reference golden: cells entered in the order idx 2, 0, 1 = 30, 10, 20 -> first = 30, last = 20."""
    case = KAT[name]
    size, keys, vals = dec_store(case["in"])
    g = pkg.HipStore(size, case["type"], 0.0).track_order()
    for k, v in zip(keys, vals):
        g.set_value(int(k), float(v))
    assert g.order_tracked == 2 and g.keys().tolist() == [2, 0, 1]
    out = g.drill_up(case["oldLen"], case["newLen"], case["maps"], case["method"])
    _, okeys, ovals = dec_store(case["out"])
    assert out.get_data_f64().tolist() == ovals.tolist() and out.keys().tolist() == okeys.tolist()


def test_set_value_delete_and_readd_moves_a_key_to_the_end():
    o, g = pair(6)
    for k, v in ((0, 1.0), (1, 2.0), (2, 3.0), (4, 5.0)):
        o.set(k, v)
        g.set_value(k, v)
    assert g.order_tracked == 1  # still ascending: no sidecar yet
    for k, v in ((1, 0.0), (1, 7.0), (3, 9.0), (0, 8.0)):  # delete 1, re-add it (moves to the end), add 3, overwrite 0 (stays first)
        o.set(k, v)
        g.set_value(k, v)
    assert o.entries()[0].tolist() == [0, 2, 4, 1, 3]
    same(o, g, "after setValue")
    m = [np.zeros(6, np.uint32)]
    for method, want in (("first", 8.0), ("last", 9.0)):
        assert g.drill_up([6], [1], m, method).get_data_f64().tolist() == [want] == o.drill_up([6], [1], m, method).dense()[0].tolist()


def test_bulk_write_over_a_partly_filled_store():
    """`data=` keeps the place of cells that stay set and appends the new ones in index order (in-memory.js:39-46)."""
    o, g = pair(8, "float32", float("nan"))
    for k in (5, 6):
        o.set(k, 1.0 + k)
        g.set_value(k, 1.0 + k)
    data = np.array([1, np.nan, 3, 4, np.nan, 50, np.nan, 8.0])
    o.set_data(data)
    g.set_data_f64(data)
    assert o.entries()[0].tolist() == [5, 0, 2, 3, 7]
    same(o, g, "data= over a partly filled store")
    o.fill(2.5)
    g.fill(2.5)
    same(o, g, "fill")


def test_sparse_rollup_chain_first_and_last():
    """The chain of tests/test_typed_storage_differences.py::test_sparse_rollup_order_is_by_index_too, tracked: the
    roll-up of the outer dimension leaves the result in first-hit order [1, 0]; `first` then yields column 1."""
    vals = np.array([0.0, 5.0, 7.0, 11.0])
    m0 = [np.zeros(2, np.uint32), ident(2)]
    m1 = [np.zeros(1, np.uint32), np.zeros(2, np.uint32)]
    o, g = pair(4)
    o.set_data(vals)
    g.set_data_f64(vals)
    o1, g1 = o.drill_up([2, 2], [1, 2], m0, "sum"), g.drill_up([2, 2], [1, 2], m0, "sum")
    same(o1, g1, "sum over dim0")
    assert g1.keys().tolist() == [1, 0]
    for method, want in (("first", 16.0), ("last", 7.0)):
        assert g1.drill_up([1, 2], [1, 1], m1, method).get_data_f64().tolist() == [want]
        assert o1.drill_up([1, 2], [1, 1], m1, method).dense()[0].tolist() == [want]


@pytest.mark.parametrize("seed", range(30))
def test_random_operation_chains_keep_the_reference_order(seed):
    """Random stores filled out of order, then chains of drillUp / dice / reorder / drillDown / load: values, masks
    and key order against the oracle after every step."""
    rng = np.random.default_rng(500 + seed)
    type_name, default = [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)][seed % 5]
    lens = [int(x) for x in rng.integers(2, 6, size=3)]
    n = int(np.prod(lens))
    o, g = pair(n, type_name, default)
    for k in rng.permutation(n)[: max(2, int(n * 0.7))]:
        v = float(rng.integers(1, 40))
        o.set(int(k), v)
        g.set_value(int(k), v)
    same(o, g, "filled out of order")
    for step in range(4):
        op = ["drillUp", "dice", "reorder", "drillDown"][int(rng.integers(0, 4))]
        if op == "drillUp":
            d = int(rng.integers(0, len(lens)))
            G = int(rng.integers(1, lens[d] + 1))
            gmap = rng.integers(0, G, size=lens[d]).astype(np.uint32)
            new = list(lens)
            new[d] = G
            maps = [gmap if i == d else ident(l) for i, l in enumerate(lens)]
            method = ["first", "last", "sum", "highest", "first", "last"][int(rng.integers(0, 6))]
            o, g = o.drill_up(lens, new, maps, method), g.drill_up(lens, new, maps, method)
            if method in ("sum",):  # accumulated in index order here, insertion order there: identical for these small integers
                pass
            lens = new
        elif op == "dice":
            sel = [rng.permutation(l)[: int(rng.integers(1, l + 1))].astype(np.int32) for l in lens]
            new = [len(s) for s in sel]
            o, g = o.dice(lens, new, sel), g.dice(lens, new, sel)
            lens = new
        elif op == "reorder":
            perm = [int(x) for x in rng.permutation(len(lens))]
            o, g = o.reorder(lens, perm), g.reorder(lens, perm)
            lens = [lens[p] for p in perm]
        else:
            d = int(rng.integers(0, len(lens)))
            child = np.repeat(np.arange(lens[d]), 2).astype(np.uint32)
            new = list(lens)
            new[d] = lens[d] * 2
            maps = [child if i == d else ident(l) for i, l in enumerate(lens)]
            o, g = o.drill_down(lens, new, maps, "first"), g.drill_down(lens, new, maps, "first")
            lens = new
        # the typed store holds rounded values: continue the oracle from them, keeping ITS key order
        ev, es = expected_typed(o)
        keys = o.entries()[0]
        o2 = OracleStore(o.size, type_name, default)
        for k in keys:
            if es[int(k)] == 2:
                o2.set(int(k), float(ev[int(k)]))
        o = o2
        assert g.order_tracked >= 1
        same(o, g, "%s (step %d)" % (op, step))


def fill_in_order(n, type_name, default, order, values):
    o, g = pair(n, type_name, default)
    for k in order:
        o.set(int(k), float(values[int(k)]))
        g.set_value(int(k), float(values[int(k)]))
    return o, g


def test_sums_add_in_insertion_order_not_index_order():
    """float64 addition is not associative: [1e16, 1, -1e16] entered as cells 0, 2, 1 sums to 1 in the reference
    (1e16 - 1e16 = 0 drops the key, the 1 brings it back) and to 0 in index order (1 is absorbed by 1e16 first)."""
    vals = np.array([1e16, 1.0, -1e16, 4.0, 5.0, 0.0])
    order = [0, 2, 1, 4, 3]
    m = [np.array([0, 0, 0, 1, 1, 1], np.uint32)]
    for method in ("sum", "average", "product"):
        o, g = fill_in_order(6, "float64", 0.0, order, vals)
        oo, gg = o.drill_up([6], [2], m, method), g.drill_up([6], [2], m, method)
        same(oo, gg, method)
    o, g = fill_in_order(6, "float64", 0.0, order, vals)
    got = g.drill_up([6], [2], m, "sum")
    assert got.get_data_f64().tolist() == [1.0, 9.0] and got.keys().tolist() == [0, 1]
    # the same cells entered ascending: the ordinary kernels (index order = insertion order): 1e16 + 1 = 1e16, - 1e16 = 0,
    # the key is dropped (in-memory.js:126-131) and never comes back
    o, g = fill_in_order(6, "float64", 0.0, [0, 1, 2, 3, 4], vals)
    same(o.drill_up([6], [2], m, "sum"), g.drill_up([6], [2], m, "sum"), "ascending")
    got = g.drill_up([6], [2], m, "sum")
    assert got.get_data_f64().tolist() == [0.0, 9.0] and got.keys().tolist() == [1]


def test_a_key_dropped_by_a_running_default_re_enters_at_the_end():
    """sum of [3, -3, 7] in that order: the key is deleted when the running sum hits 0 and re-inserted by the 7 — behind
    the other output cell, whose first contribution came earlier (in-memory.js:311-318)."""
    vals = np.array([3.0, 10.0, -3.0, 7.0])
    m = [np.array([0, 1, 0, 0], np.uint32)]
    o, g = fill_in_order(4, "float32", 0.0, [0, 2, 1, 3], vals)
    oo, gg = o.drill_up([4], [2], m, "sum"), g.drill_up([4], [2], m, "sum")
    assert oo.entries()[0].tolist() == [1, 0]
    same(oo, gg, "drop and re-enter")
    # `first` / `last` of that result see the order
    m1 = [np.zeros(2, np.uint32)]
    assert gg.drill_up([2], [1], m1, "first").get_data_f64().tolist() == [10.0] == oo.drill_up([2], [1], m1, "first").dense()[0].tolist()


@pytest.mark.parametrize("seed", range(24))
def test_replayed_rollups_random(seed):
    """Stores filled in random order with values that cancel, absorb and hit the default; every rule, one or several
    rolled-up dimensions (first / last over several at once included): values, mask and key order against the oracle."""
    rng = np.random.default_rng(9100 + seed)
    type_name, default = [("float64", 0.0), ("float32", 0.0), ("float64", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float32", float("nan"))][seed % 6]
    lens = [int(x) for x in rng.integers(2, 7, size=3)]
    n = int(np.prod(lens))
    # (2^54 absorbs 1 in float64 and, like every value here, is exact in float32 too)
    pool = np.array([1.0, -1.0, 2.0, -2.0, 3.0, 0.5, 2.0 ** 54, -2.0 ** 54, 2.0 ** -10, 7.0]) if type_name.startswith("float") else np.array([1.0, 2.0, 3.0, 5.0, 7.0, 11.0])
    if type_name == "int32":
        pool = np.concatenate([pool, -pool])
    vals = rng.choice(pool, size=n)
    order = rng.permutation(n)[: max(2, int(n * 0.8))]
    several = seed % 3 == 0
    maps, new = [], []
    for d, l in enumerate(lens):
        if d == seed % 3 or several:
            G = int(rng.integers(1, l + 1))
            gm = rng.integers(0, G, size=l).astype(np.uint32)
            gm[rng.integers(0, l)] = G - 1  # every new item is named by the extents, not necessarily by a member
            maps.append(gm)
            new.append(G)
        else:
            maps.append(ident(l))
            new.append(l)
    for method in ("sum", "average", "product", "first", "last", "highest", "lowest"):
        o, g = fill_in_order(n, type_name, default, order, vals)
        same(o.drill_up(lens, new, maps, method), g.drill_up(lens, new, maps, method), "%s %s %s" % (method, lens, new))


def test_load_appends_in_the_other_stores_index_order():
    my_len, his_len = [3, 4], [2, 3]
    o, g = pair(12, "float32", float("nan"))
    for k, v in ((7, 1.0), (2, 2.0), (11, 3.0)):
        o.set(k, v)
        g.set_value(k, v)
    ho, hg = pair(6, "float32", 0.0)
    for k, v in ((4, 40.0), (0, 10.0), (3, 30.0)):
        ho.set(k, v)
        hg.set_value(k, v)
    h2m = [np.array([2, 0], np.int32), np.array([3, 1, 0], np.int32)]
    o.load(ho, my_len, his_len, h2m)
    g.load(hg, my_len, his_len, h2m)
    same(o, g, "load")
    # into a fresh tracked store with a monotone remap: still ascending, no sidecar
    o2, g2 = pair(12, "float32", 0.0)
    h2m2 = [np.array([0, 2], np.int32), np.array([0, 1, 3], np.int32)]
    o2.load(ho, my_len, his_len, h2m2)
    g2.load(hg, my_len, his_len, h2m2)
    same(o2, g2, "load into a fresh store")
    assert g2.order_tracked == 1


def test_sparse_form_round_trip_keeps_the_order():
    o, g = pair(9, "int32", float("nan"))
    for k, v in ((8, 1), (3, 0), (5, -2), (0, 7)):
        o.set(k, float(v))
        g.set_value(k, float(v))
    idx, vals = g.to_sparse()
    assert idx.tolist() == [8, 3, 5, 0] and vals.tolist() == [1, 0, -2, 7]
    back = pkg.HipStore.from_sparse(9, "int32", float("nan"), idx, vals)  # a blob in another order carries that order
    assert back.order_tracked == 2 and back.keys().tolist() == [8, 3, 5, 0]
    m = [np.zeros(9, np.uint32)]
    assert back.drill_up([9], [1], m, "first").get_data_f64().tolist() == [1.0] and back.drill_up([9], [1], m, "last").get_data_f64().tolist() == [7.0]
    with pytest.raises(pkg.OlapError, match="ordered:"):
        back.totals([9], ["first"])
    # an untracked store keeps answering by flat index
    plain = pkg.HipStore(9, "int32", float("nan"))
    for k, v in ((8, 1), (3, 0), (5, -2), (0, 7)):
        plain.set_value(k, float(v))
    assert plain.order_tracked == 0 and plain.keys().tolist() == [0, 3, 5, 8]
    assert plain.drill_up([9], [1], m, "first").get_data_f64().tolist() == [7.0]
