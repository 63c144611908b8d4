"""Pins the CPU oracle (oracle/olap_oracle.c) to the reference.

Every vector under tests/golden/ was produced by executing the reference's own
src/store/in-memory.js + src/dimension/generic.js (oracle/gen_golden.js); the oracle must
reproduce keys (Map insertion order) and float64 values bit for bit.
"""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, config_cube, dec_num, dec_store, default_of, load_cases, same_f64
from oracle.oracle import OracleStore

KAT = load_cases("store_kat.json")
RND = load_cases("store_random.json")
CFG = load_cases("configs.json")


def build_store(case, key="in", default_key="default"):
    size, keys, vals = dec_store(case[key])
    s = OracleStore(size, case["type"], default_of(case, default_key))
    if "iota" in case[key]:
        s.fill(vals[0])
    else:
        for k, v in zip(keys, vals):
            s.set(int(k), float(v))
    k2, v2 = s.entries()
    assert np.array_equal(k2, keys) and same_f64(v2, vals), "oracle setValue diverges from the reference"
    return s


def run_case(case):
    s = build_store(case)
    op = case["op"]
    if op == "drillUp":
        return s.drill_up(case["oldLen"], case["newLen"], case["maps"], case["method"])
    if op == "drillDown":
        dist = None
        if case.get("distributions") is not None:
            dist = [dec_num(x) for x in case["distributions"]]
        return s.drill_down(case["oldLen"], case["newLen"], case["maps"], case["method"], dist)
    if op == "dice":
        return s.dice(case["oldLen"], case["newLen"], case["sel"])
    if op == "reorder":
        return s.reorder(case["oldLen"], case["perm"])
    if op == "load":
        his = build_store(case, "his", "hisDefault")
        s.load(his, case["myLen"], case["hisLen"], case["hisToMine"])
        return s
    raise AssertionError(op)


@pytest.mark.parametrize("case", KAT + RND, ids=lambda c: c["name"])
def test_oracle_matches_reference(case):
    if "throws" in case:
        with pytest.raises(ValueError, match=case["throws"]):
            run_case(case)
        return
    out = run_case(case)
    size, keys, vals = dec_store(case["out"])
    assert out.size == size
    k, v = out.entries()
    assert np.array_equal(k, keys), f"key set / insertion order differs: {k[:10]} vs {keys[:10]}"
    assert same_f64(v, vals), f"values differ: {v[:10]} vs {vals[:10]}"


def test_unknown_method_rejected():
    s = OracleStore(4, "float32", 0.0)
    with pytest.raises(ValueError, match="Unsupported aggregation method: median"):
        s.drill_up([4], [1], [[0, 0, 0, 0]], "median")


def test_constructor_errors():
    with pytest.raises(ValueError, match="only NaN and 0"):
        OracleStore(4, "float32", 1.0)
    with pytest.raises(ValueError, match="Invalid type"):
        OracleStore(4, "float16", 0.0)
    s = OracleStore(4, "float32", 0.0)
    with pytest.raises(ValueError, match="value length is invalid: 4 !== 3"):
        s.set_data([1, 2, 3])


@pytest.mark.parametrize("case", CFG, ids=lambda c: c["name"])
def test_oracle_configs(case):
    """BASELINE.json configs 1 and 2 (10^3 and 10^6 cells) against the reference's outputs."""
    lens, axis = case["lens"], case["axis"]
    n = int(np.prod(lens))
    s = OracleStore(n, "float32", 0.0)
    s.fill_seeded(case["seed"], case["frac"])
    # the numpy generator used by the GPU tests must be the same stream
    vals, keep = config_cube(n, case["seed"], case["frac"])
    dv, dp = s.dense()
    assert np.array_equal(dp, keep) and np.array_equal(dv.astype(np.float32), vals)
    new_len = list(lens)
    new_len[axis] = 1
    maps = [np.zeros(l, dtype=np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    out = s.drill_up(lens, new_len, maps, case["method"])
    ov, op = out.dense()
    assert out.num_keys == case["outKeys"]
    if "out" in case:
        assert same_f64(ov, [dec_num(x) for x in case["out"]])
    else:
        ref32 = np.fromfile(os.path.join(GOLDEN, case["name"] + ".f32"), dtype=np.float32)
        refp = np.fromfile(os.path.join(GOLDEN, case["name"] + ".present.u8"), dtype=np.uint8).astype(bool)
        assert np.array_equal(op, refp)
        assert np.array_equal(ov.astype(np.float32), ref32)
        for i, v in case["spots"]:
            assert same_f64([ov[i]], [dec_num(v)])
        assert s.num_keys == case["inKeys"]
