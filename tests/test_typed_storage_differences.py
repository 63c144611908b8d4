"""Where this store's results DIFFER from the reference's getData(), pinned as assertions instead of being
normalised away by golden_util.expected_typed().

The reference keeps every cell as a float64 JS number until serialize() (in-memory.js:118-133 never coerces;
:77-92 does, at serialisation) and iterates its Map in insertion order (:298).  The C ABI stores cells in the
type a store is created with, after every operation, and orders cells by flat index —
which is what an Int32 / Uint32 / Float32 STORE shows below.  (The Node host holds int32 / uint32 MEASURES in
float64 cells and so has none of the integer differences: last test here, and tests/js/gpu_test.js.)  Each test states: reference golden = X (tests/golden/store_kat.json, produced by running the
reference), this store = Y, because ...  A change to either side of these numbers must be deliberate."""
import numpy as np
import pytest

from conftest import load_package
from golden_util import dec_store, load_cases
from oracle.oracle import OracleStore

pytestmark = pytest.mark.gpu

pkg = load_package()
KAT = {c["name"]: c for c in load_cases("store_kat.json")}


def run_case(case, ascending):
    """The golden case on the typed store; cells are entered one by one in the golden's insertion order."""
    size, keys, vals = dec_store(case["in"])
    g = pkg.HipStore(size, case["type"], 0.0)
    order = np.argsort(keys) if ascending else np.arange(len(keys))
    for i in order:
        g.set_value(int(keys[i]), float(vals[i]))
    out = g.drill_up(case["oldLen"], case["newLen"], case["maps"], case["method"])
    return out.get_data_f64(), out.get_status()


def golden_out(case):
    size, keys, vals = dec_store(case["out"])
    dense = np.zeros(size)
    dense[keys.astype(np.int64)] = vals
    return dense


def test_int32_average_keeps_no_fraction():
    """reference golden = 7.5 (a float64 in the Map; getData() returns it), this store = 7: the Int32 cell holds
    ToInt32(7.5) — the value the reference itself would write at serialize() (in-memory.js:84)."""
    case = KAT["int32_average_fraction"]
    assert golden_out(case).tolist() == [7.5]
    got, st = run_case(case, ascending=True)
    assert got.tolist() == [7.0] and st.tolist() == [2]


def test_uint32_sum_wraps_at_storage_not_in_the_sum():
    """reference golden = 8000000000 (float64, no 32-bit wrap), this store = 8000000000 mod 2^32 = 3705032704: the sum
    itself is float64 here too (no wrap inside the accumulation), the Uint32 cell holds ToUint32 of it."""
    case = KAT["uint32_sum_no_wrap"]
    assert golden_out(case).tolist() == [8000000000.0]
    got, st = run_case(case, ascending=True)
    assert got.tolist() == [8000000000.0 % 4294967296.0] and st.tolist() == [2]


def test_chained_operations_round_after_every_step():
    """average of [1, 2] then sum of two such cells, Int32: reference = 1.5 + 1.5 = 3, this store = 1 + 1 = 2
    (each intermediate store is typed).  Float32: 0.1 + 0.2 differs from the float64 chain by < 1e-7 relative."""
    lens, mid, new = [2, 2], [2, 1], [1, 1]
    m1 = [np.arange(2, dtype=np.uint32), np.zeros(2, np.uint32)]
    m2 = [np.zeros(2, np.uint32), np.zeros(1, np.uint32)]
    o = OracleStore(4, "int32", 0.0)
    o.set_data(np.array([1, 2, 1, 2], np.float64))
    ref = o.drill_up(lens, mid, m1, "average").drill_up(mid, new, m2, "sum").dense()[0]
    g = pkg.HipStore(4, "int32", 0.0)
    g.set_data_f64([1, 2, 1, 2])
    got = g.drill_up(lens, mid, m1, "average").drill_up(mid, new, m2, "sum").get_data_f64()
    assert ref.tolist() == [3.0] and got.tolist() == [2.0]
    o = OracleStore(2, "float32", 0.0)
    o.set_data(np.array([0.1, 0.2]))
    ref = o.drill_up([2], [1], [np.zeros(2, np.uint32)], "sum").dense()[0][0]
    g = pkg.HipStore(2, "float32", 0.0)
    g.set_data_f64([0.1, 0.2])
    got = g.drill_up([2], [1], [np.zeros(2, np.uint32)], "sum").get_data_f64()[0]
    assert ref == 0.1 + 0.2 and got == float(np.float32(np.float64(np.float32(0.1)) + np.float64(np.float32(0.2))))
    assert got != ref and abs(got - ref) <= 1e-7 * ref


def test_result_that_rounds_to_the_default_becomes_unset():
    """Int32 over a 0 default, average of [1, -2] = -0.5: reference = a set cell holding -0.5, this store = unset
    (ToInt32(-0.5) = 0 is the default, and a typed cell equal to the default cannot be told from an unset one —
    the same rule setValue applies to a literal 0, in-memory.js:126-131)."""
    o = OracleStore(2, "int32", 0.0)
    o.set_data(np.array([1.0, -2.0]))
    r = o.drill_up([2], [1], [np.zeros(2, np.uint32)], "average")
    assert r.dense()[0].tolist() == [-0.5] and r.num_keys == 1
    g = pkg.HipStore(2, "int32", 0.0)
    g.set_data_f64([1, -2])
    out = g.drill_up([2], [1], [np.zeros(2, np.uint32)], "average")
    assert out.get_data_f64().tolist() == [0.0] and out.get_status().tolist() == [0]


@pytest.mark.parametrize("name,reference,by_index", [("first_insertion_order", 30.0, 10.0), ("last_insertion_order", 20.0, 30.0)])
def test_first_and_last_use_the_flat_index_not_the_insertion_order(name, reference, by_index):
    """Cells entered in the order idx 2, 0, 1 = 30, 10, 20.  reference golden: first = 30, last = 20 (Map insertion
    order, in-memory.js:298).  This store = 10 / 30: a dense buffer has no insertion order, `first` / `last` follow the
    ascending flat index — the reference's own order whenever a store was filled ascending (`data=`, `fill`) and was
    dense when it was rolled up.  Entered ascending, the reference gives 10 / 30 as well (checked through the oracle)."""
    case = KAT[name]
    assert golden_out(case).tolist() == [reference]
    for ascending in (False, True):
        got, st = run_case(case, ascending)
        assert got.tolist() == [by_index] and st.tolist() == [2]
    o = OracleStore(3, "float32", 0.0)
    for k, v in ((0, 10.0), (1, 20.0), (2, 30.0)):
        o.set(k, v)
    assert o.drill_up(case["oldLen"], case["newLen"], case["maps"], case["method"]).dense()[0].tolist() == [by_index]


def test_sparse_rollup_order_is_by_index_too():
    """A chain the reference resolves by FIRST-HIT order: cube [2, 2] with cell (0,0) unset; drillUp(dim0 -> all) visits
    (0,1), (1,0), (1,1), so its result Map is ordered [1, 0]; a following drillUp(dim1 -> all, 'first') therefore
    yields the value of column 1.  This store answers with column 0 (ascending index).  Only `first` / `last` over a
    SPARSE store that an earlier roll-up of an outer dimension produced can see this."""
    vals = np.array([0.0, 5.0, 7.0, 11.0])  # (0,0) unset under a 0 default
    m0 = [np.zeros(2, np.uint32), np.arange(2, dtype=np.uint32)]
    m1 = [np.zeros(1, np.uint32), np.zeros(2, np.uint32)]
    o = OracleStore(4, "float32", 0.0)
    o.set_data(vals)
    step1 = o.drill_up([2, 2], [1, 2], m0, "sum")
    assert step1.entries()[0].tolist() == [1, 0]  # insertion order of the result
    assert step1.drill_up([1, 2], [1, 1], m1, "first").dense()[0].tolist() == [16.0]
    g = pkg.HipStore(4, "float32", 0.0)
    g.set_data_f64(vals)
    got = g.drill_up([2, 2], [1, 2], m0, "sum").drill_up([1, 2], [1, 1], m1, "first").get_data_f64()
    assert got.tolist() == [7.0]


@pytest.mark.parametrize("declared", ["int32", "uint32"])
@pytest.mark.parametrize("default", [0.0, float("nan")])
def test_integer_measure_in_float64_cells_equals_the_reference_exactly(declared, default):
    """The way out of the differences above, and what the Node host does by default (js/store/hip.js cellTypeOf):
    an int32 / uint32 MEASURE held in float64 CELLS.  Every operation then works on the numbers the reference's
    Map holds; the one place the declared type matters before serialize() is drillDown's remainder rule
    (in-memory.js:343, :403-417), asked for with OLAP_DRILLDOWN_INTEGER_MEASURE.  Compared with the oracle of the
    DECLARED type, bit for bit, without expected_typed()'s coercion."""
    rng = np.random.default_rng(5)
    # fractions, values past 2^32 and negative values in a uint32 measure: all legal numbers of the reference's Map
    for child_map, inner in [(np.repeat(np.arange(4), [3, 1, 5, 3]), 520), (np.arange(12) % 4, 3), (np.repeat(np.arange(2), [19, 2]), 2056)]:
        G, K = int(child_map.max()) + 1, len(child_map)
        old_len, new_len = [3, G, inner], [3, K, inner]
        n_old = int(np.prod(old_len))
        vals = rng.integers(-40, 90, size=n_old).astype(np.float64) * 0.5
        vals[rng.random(n_old) < 0.05] = 6e9 + 0.25
        dense = np.where(rng.random(n_old) < 0.25, default, vals)
        maps = [np.arange(3, dtype=np.uint32), child_map.astype(np.uint32), np.arange(inner, dtype=np.uint32)]
        for method in ("sum", "average"):
            o = OracleStore(n_old, declared, default)
            o.set_data(dense)
            want = o.drill_down(old_len, new_len, maps, method)
            g = pkg.HipStore(n_old, "float64", default)
            g.set_data_f64(dense)
            out = g.drill_down(old_len, new_len, maps, method, integer_measure=True)
            vals, present = want.dense()
            assert np.array_equal(out.get_data_f64(), np.where(present, vals, default), equal_nan=True), (declared, method, inner)
            assert np.array_equal(out.get_status() != 0, present)
            if method == "sum":  # not what a float64 MEASURE gives (plain division)
                plain = g.drill_down(old_len, new_len, maps, method).get_data_f64()
                assert not np.array_equal(plain, out.get_data_f64(), equal_nan=True)
    # and back up: average keeps its fraction, sums pass 2^32
    o = OracleStore(4, declared, default)
    o.set_data(np.array([7.0, 8.0, 4e9, 4e9]))
    up_maps = [np.array([0, 0, 1, 1], np.uint32)]
    g = pkg.HipStore(4, "float64", default)
    g.set_data_f64([7.0, 8.0, 4e9, 4e9])
    assert g.drill_up([4], [2], up_maps, "average").get_data_f64().tolist() == [7.5, 4e9]
    assert g.drill_up([4], [2], up_maps, "sum").get_data_f64().tolist() == o.drill_up([4], [2], up_maps, "sum").dense()[0].tolist() == [15.0, 8e9]
