"""Worker for tests/test_sharded_gloo.py::test_sharded_single_process (GPU box, one process).

1. direct transport — olap_comm_init_all({0,0,0}) and {0,0}: three / two ranks on cuda:0, the whole
   sharded store API and olap_shard_drillup_step (events, both depths, every placement) end to end;
2. RCCL — one-rank communicators from olap_comm_init_all({0}) and olap_comm_init_rank: every collective
   the exchange issues (ReduceScatter / AllReduce / Reduce / AllGather, Broadcast in gather) really runs
   through librccl on a device buffer.
Expectations: the CPU oracle on the whole cube."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from golden_util import config_cube, expected_typed  # noqa: E402
from oracle.oracle import OracleStore  # noqa: E402
from sharded_cases import CASES, case_data, methods_of  # noqa: E402

pkg = load_package()
from olap_in_memory_amd import capi  # noqa: E402
from olap_in_memory_amd.sharded import Comm, ShardedStore  # noqa: E402

capi.check(capi.lib().olap_set_device(0))


def oracle_store(case):
    o = OracleStore(int(np.prod(case["lens"])), case["dtype"], case["default"])
    o.set_data(case_data(case))
    return o


def close(method, dtype, got, exp):
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    if method == "product" and dtype.startswith("float"):  # gathered partial products are rounded per rank
        return np.allclose(got, exp, rtol=1e-5, atol=0, equal_nan=True)
    # sum / average: float64 partials rounded once, and case data whose float64 sums are exact in any order -> bit for bit
    return np.array_equal(got, exp, equal_nan=True)


def run_steps(comm, what):
    world = comm.world
    for name, case in CASES.items():
        lens, dtype, default = case["lens"], case["dtype"], case["default"]
        o = oracle_store(case)
        s = ShardedStore(comm, lens, dtype, default).set_data_f64(case_data(case))
        want = expected_typed(o)[0].astype(np.float64)
        if default != default and dtype in ("int32", "uint32"):  # getValue of an unset integer cell under a NaN default: NaN (:118-120)
            want = np.where(expected_typed(o)[1] == 2, want, np.nan)
        assert np.array_equal(s.get_data_f64(), want, equal_nan=True), name
        assert np.array_equal(s.get_status(), expected_typed(o)[1]), name
        maps = [np.asarray(case["row_map"], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in lens[1:]]
        new_len = [case["groups"]] + lens[1:]
        n_out = int(np.prod(new_len))
        vals, stat = s.step_inputs()
        for method in methods_of(case):
            ev, es = expected_typed(o.drill_up(lens, new_len, maps, method))
            for placement in (capi.PLACE_SCATTER, capi.PLACE_ALL, capi.PLACE_ROOT):
                for depth in (1, 2):
                    op = s.plan_drillup_dim0(case["row_map"], case["groups"], method, placement=placement, depth=depth)
                    for _ in range(3 if depth == 2 else 1):  # with two buffer sets: reuse of a set behind its previous exchange
                        op.step(vals, stat)
                    op.wait()
                    gv, gs = np.full(n_out, np.nan), np.full(n_out, -1, np.int64)
                    for i in range(comm.local_count):
                        v, st, first = op.result_host(i)
                        if op.placement != capi.PLACE_SCATTER and v.size:
                            assert first == 0 and v.size == n_out
                        if st is None:
                            st = np.where(v != 0, 2, 0)
                        gv[first:first + v.size] = v
                        gs[first:first + v.size] = st
                    tag = "%s %s %s placement %d depth %d" % (what, name, method, placement, depth)
                    assert np.array_equal(gs, es), tag + " (mask)"
                    assert close(method, dtype, gv, ev), tag
                    if op.placement == capi.PLACE_ROOT and world > 1:
                        assert op.result(1)[3] == 0, tag
                    op.destroy()
            # the store-level operation the Node host binds: with a new row per rank the result stays sharded along the
            # new leading dimension (rank r keeps rows [r*p, (r+1)*p), p = ceil(G / world)), otherwise it arrives whole
            res = s.drill_up(new_len, maps, method)
            if case["groups"] >= world:
                assert isinstance(res, ShardedStore) and res.size == n_out
                p_rows = -(-case["groups"] // world)
                assert res.bounds == [min(r * p_rows, case["groups"]) for r in range(world + 1)], res.bounds
                got_v, got_s = res.get_data_f64(), res.get_status()
                if default != default and dtype in ("int32", "uint32"):
                    got_v = np.where(got_s == 2, got_v, 0.0)
                # and it is a sharded store like any other: roll the rest of dimension 0 up from there
                if method == "sum" and name == "f32_zero":
                    again = res.drill_up([1] + new_len[1:], [np.zeros(case["groups"], np.uint32)] + maps[1:], "sum")
                    e2, _ = expected_typed(o.drill_up(lens, [1] + new_len[1:], [np.zeros(lens[0], np.uint32)] + maps[1:], "sum"))
                    assert isinstance(again, ShardedStore if world == 1 else pkg.HipStore)  # one row left: whole, unless there is one rank
                    assert np.allclose(again.get_data_f64() if world == 1 else again.get_data(), e2, rtol=1e-5, atol=0)
            else:
                assert isinstance(res, pkg.HipStore) and res.size == n_out
                got_v, got_s = res.get_data().astype(np.float64), res.get_status()
            assert np.array_equal(got_s, es), "%s %s %s store" % (what, name, method)
            assert close(method, dtype, got_v, ev), "%s %s %s store" % (what, name, method)
        del s


def run_store_api(comm, what):
    lens = [7, 6, 10]
    case = CASES["f32_nan"]
    o = oracle_store(case)
    s = ShardedStore(comm, lens, "float32", float("nan")).set_data_f64(case_data(case))
    ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731

    def same(got, exp, tag):
        ev, es = expected_typed(exp)
        if isinstance(got, ShardedStore):
            gv, gs = got.get_data_f64(), got.get_status()
        else:
            gv, gs = got.get_data().astype(np.float64), got.get_status()
        assert np.array_equal(gv, ev.astype(np.float64), equal_nan=True) and np.array_equal(gs, es), what + " " + tag

    same(s.gather(), o, "gather")
    same(ShardedStore.scatter(comm, s.gather(), lens), o, "scatter")
    tracked = s.gather()
    tracked.track_order()
    try:  # a store that tracks the reference Map's insertion order cannot be split by rows without losing it
        ShardedStore.scatter(comm, tracked, lens)
        raise AssertionError("expected an 'ordered:' refusal")
    except capi.OlapError as e:
        assert str(e).startswith("ordered:") and "one device" in str(e), str(e)
    same(s.clone(), o, "clone")
    assert abs(s.total - o.total()) < 1e-9
    same(s.drill_up([7, 6, 2], [ident(7), ident(6), (np.arange(10) % 2).astype(np.uint32)], "average"),
         o.drill_up(lens, [7, 6, 2], [ident(7), ident(6), (np.arange(10) % 2).astype(np.uint32)], "average"), "drillUp axis 2")
    sel = [np.arange(7), [4, 0, -1, 2], [9, 8, 1]]
    same(s.dice([7, 4, 3], sel), o.dice(lens, [7, 4, 3], sel), "dice axes 1, 2")
    rows = [np.array([1, 2, 4, 6]), np.arange(6), np.arange(10)]
    picked = s.dice([4, 6, 10], rows)
    same(picked, o.dice(lens, [4, 6, 10], rows), "dice rows")
    assert picked.bounds == [int(np.searchsorted([1, 2, 4, 6], b)) for b in s.bounds]
    m = [np.zeros(4, np.uint32), ident(6), ident(10)]
    same(picked.drill_up([1, 6, 10], m, "last"), o.dice(lens, [4, 6, 10], rows).drill_up([4, 6, 10], [1, 6, 10], m, "last"), "rows then dim0 last")
    pairs = [(np.arange(7) // 2).astype(np.uint32), ident(6), ident(10)]  # 7 rows -> 4 groups: stays sharded, uneven blocks
    for method in ("sum", "average", "highest", "last"):
        same(s.drill_up([4, 6, 10], pairs, method), o.drill_up(lens, [4, 6, 10], pairs, method), "dim0 -> 4 groups " + method)
    dm = [ident(7), ident(6), np.repeat(np.arange(10), 3).astype(np.uint32)]
    same(s.drill_down([7, 6, 30], dm, "sum"), o.drill_down(lens, [7, 6, 30], dm, "sum"), "drillDown axis 2")
    same(s.reorder([0, 2, 1]), o.reorder(lens, [0, 2, 1]), "reorder")
    for bad, args in (("reorder", ([2, 1, 0],)), ("dice", ([2, 6, 10], [[3, 1], np.arange(6), np.arange(10)])),
                      ("drill_down", ([14, 6, 10], [np.repeat(np.arange(7), 2), ident(6), ident(10)], "sum"))):
        try:
            getattr(s, bad)(*args)
            raise AssertionError("expected a 'sharded:' refusal of " + bad)
        except capi.OlapError as e:
            assert "sharded:" in str(e), str(e)
    # single cells are routed to the rank that owns the row
    for idx in (0, 59, 60, 239, 240, 419):
        assert np.array_equal(np.float64(s.get_value(idx)[0]), np.float64(o.get(idx)), equal_nan=True)
    s.set_value(245, 12.5)
    s.set_value(3, None)
    o.set(245, 12.5)
    o.set(3, None)
    same(s, o, "setValue")
    s.fill(2.0)
    o.fill(2.0)
    same(s, o, "fill")
    # the synthetic generator: every rank fills its own slab of ONE global stream
    g = ShardedStore(comm, [9, 50], "float32", float("nan")).fill_seeded(5, 0.6)
    v, keep = config_cube(450, 5, 0.6)
    assert np.array_equal(g.get_data_f64(), np.where(keep, v.astype(np.float64), np.nan), equal_nan=True), what + " fill_seeded"


for devices in ([0, 0, 0], [0, 0]):
    c = Comm.init_all(devices)
    assert c.transport == "direct" and c.world == len(devices) and c.local_count == len(devices)
    run_steps(c, "direct%d" % len(devices))
    run_store_api(c, "direct%d" % len(devices))
    c.destroy()
print("direct transport ok", flush=True)

# The same ranks with one ISSUING THREAD per rank (what a process that drives several devices over RCCL uses; here
# forced onto the direct transport so that the hand-over, the barriers between the workers and the relay of a worker's
# error run on a one-GPU box), and the unfused direct path issued by the calling thread.
for env, tag in (("OLAP_SHARD_THREADS", "threads"), ("OLAP_SHARD_NO_FUSED", "unfused")):
    os.environ[env] = "1"
    for devices in ([0, 0, 0], [0, 0]):
        c = Comm.init_all(devices)
        run_steps(c, "%s%d" % (tag, len(devices)))
        run_store_api(c, "%s%d" % (tag, len(devices)))
        if env == "OLAP_SHARD_THREADS":  # a failure inside a worker reaches the caller as an ordinary error, and the workers live on
            bad = ShardedStore(c, [4, 3], "int32", float("nan")).set_data_f64(np.arange(12, dtype=np.float64))
            opb = bad.plan_drillup_dim0([0, 0, 0, 0], 1, "sum")
            vals, _ = bad.step_inputs()
            try:
                opb.step(vals, None)  # integer cells over a NaN default: the mask is required
                raise AssertionError("expected the workers to refuse a step without the mask")
            except capi.OlapError as e:
                assert "status mask is required" in str(e), str(e)
            opb.destroy()
            ok = ShardedStore(c, [4, 3], "float32", 0.0).set_data_f64(np.arange(12, dtype=np.float64))
            assert ok.drill_up([1, 3], [np.zeros(4, np.uint32), np.arange(3, dtype=np.uint32)], "sum").get_data().tolist() == [18.0, 22.0, 26.0]
        c.destroy()
    del os.environ[env]
    print("direct transport, %s ok" % tag, flush=True)

c = Comm.init_all([0])
assert c.transport == "rccl" and c.world == 1
run_steps(c, "rccl-all")
run_store_api(c, "rccl-all")
c.destroy()
c = Comm.init_rank(Comm.unique_id(), 1, 0, 0)
assert c.transport == "rccl" and c.local_rank(0) == 0
run_steps(c, "rccl-rank")
c.destroy()
try:
    Comm.init_all([0, 0, 1] if capi.lib().olap_device_count() > 1 else [0, 0, 7])
    raise AssertionError("a mixed device list must be refused")
except capi.OlapError:
    pass
print("sharded single-process ok")
