"""Multi-rank rehearsal of the dim0-sharded path (include/olap_hip.h "Multi-GPU",
olap-in-memory_amd/csrc/olap_sharded.hip) over gloo, world sizes 2 and 3.

CPU run: partition, row sub-maps and the recipe come from libolapgpu's host-only entry points, the
oracle plays the local kernels: what is under test is the MATHS of "local partial + one collective
+ finish" for every store kind (NaN / 0 default, float / integer cells, ranks without rows).
GPU runs (-m gpu): the product path itself — ranks sharing the one GPU over gloo, one process with
the direct transport, and RCCL on a one-rank communicator."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_package
from golden_util import expected_typed
from oracle.oracle import OracleStore
from sharded_cases import CASES, case_data, methods_of

HERE = os.path.dirname(os.path.abspath(__file__))
load_package()


def run_workers(engine, tmp_path, world=2):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + world), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_worker.py"), engine, out], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return [json.load(open("%s.%d" % (out, r))) for r in range(world)]


def expected_dim0(case, method):
    lens = case["lens"]
    maps = [np.asarray(case["row_map"], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in lens[1:]]
    o = OracleStore(int(np.prod(lens)), case["dtype"], case["default"])
    o.set_data(case_data(case))
    return expected_typed(o.drill_up(lens, [case["groups"]] + lens[1:], maps, method))


def assemble(results, key, n):
    """flat result cells from every rank's (first, values) piece; whole pieces must agree"""
    vals, stat = np.full(n, np.nan), np.full(n, -1, np.int64)
    for res in results:
        r = res[key]
        v, s = np.asarray(r["values"], np.float64), np.asarray(r["status"], np.int64)
        f = r["first"]
        vals[f:f + v.size] = v
        stat[f:f + s.size] = s
    return vals, stat


def check_dim0(results):
    for name, case in CASES.items():
        n_out = case["groups"] * int(np.prod(case["lens"][1:]))
        for method in methods_of(case):
            ev, es = expected_dim0(case, method)
            gv, gs = assemble(results, "%s/%s" % (name, method), n_out)
            what = "%s %s" % (name, method)
            assert np.array_equal(gs, es), what + ": status mask (must be 0 / 0x2 everywhere, OR-ed across ranks)"
            assert set(np.unique(gs)) <= {0, 2}, what
            e64 = ev.astype(np.float64)
            if method == "product" and case["dtype"].startswith("float"):
                # the gathered partial products are rounded to the cell type per rank: 1e-5 relative (north star)
                assert np.allclose(gv, e64, rtol=1e-5, atol=0, equal_nan=True), what
            else:
                # sum / average: float64 partials, added in float64, rounded ONCE — and the case data are chosen so that
                # float64 addition is exact in any order (sharded_cases.case_data): bit for bit the one-device result
                assert np.array_equal(gv, e64, equal_nan=True), what
    # the round-1 failure, spelled out
    gv, gs = assemble(results, "f32_nan_disjoint/sum", 3)
    assert gv.tolist() == [1.0, 7.0, 5.0] and gs.tolist() == [2, 2, 2]
    # the round-2 failure, spelled out: [2^24, 1 | -2^24, unset] must give 1 / set, its average 1/3 / set
    gv, gs = assemble(results, "f32_cancel/sum", 1)
    assert gv.tolist() == [1.0] and gs.tolist() == [2]
    gv, gs = assemble(results, "f32_cancel/average", 1)
    assert gv.tolist() == [float(np.float32(1.0 / 3.0))] and gs.tolist() == [2]
    gv, gs = assemble(results, "f32_cancel_nan/sum", 3)
    assert gv[0] == 1.0 and np.isnan(gv[1]) and gv[2] == 2.0 ** -30 and gs.tolist() == [2, 0, 2]
    gv, gs = assemble(results, "u32_average_overflow/average", 2)
    assert gv.tolist() == [4e9, float(np.uint32(11 / 3))] and gs.tolist() == [2, 2]
    gv, gs = assemble(results, "u32_average_overflow/sum", 2)
    assert gv.tolist() == [float(np.uint32(12e9 % 2 ** 32)), 11.0]


def test_partition_and_recipe_host_only():
    """olap_shard_bounds / olap_shard_dice_bounds / olap_shard_recipe_get need no device."""
    from olap_in_memory_amd import capi
    from olap_in_memory_amd.sharded import dice_bounds, partition_rows, recipe
    assert partition_rows(10, 8) == [0, 2, 4, 5, 6, 7, 8, 9, 10]
    assert partition_rows(320, 8) == list(range(0, 321, 40))
    assert partition_rows(3, 4) == [0, 1, 2, 3, 3]
    assert dice_bounds([0, 4, 7], [1, 2, 4, 6]) == [0, 2, 4]
    with pytest.raises(capi.OlapError, match="sharded:"):
        dice_bounds([0, 4, 7], [2, 1])
    nan = float("nan")
    f32, f64 = capi.DTYPES["float32"], capi.DTYPES["float64"]
    # Float32 sums ship the rank's float64 accumulator and are rounded once (in-memory.js:282-290 never rounds in between)
    r = recipe("float32", 0.0, "sum")
    assert (r["local_method"], r["n_payloads"], r["payload_dtype"][0], r["payload_op"][0], r["finish"], r["zero_unset"]) == \
        (capi.PARTIAL_AVERAGE, 1, f64, capi.XCHG_SUM, capi.FINISH_ROUND, False)
    r = recipe("float32", nan, "sum")  # NaN never enters an additive collective: unset partial cells ship 0, counts say who contributed
    assert (r["local_method"], r["n_payloads"], r["payload_dtype"], r["payload_op"], r["finish"], r["zero_unset"]) == \
        (capi.PARTIAL_AVERAGE, 2, [f64, capi.DTYPES["int32"]], [capi.XCHG_SUM, capi.XCHG_SUM], capi.FINISH_ROUND, False)
    r = recipe("float64", nan, "sum")  # the same for Float64 cells (rounding is the identity there)
    assert (r["local_method"], r["n_payloads"], r["payload_dtype"][0], r["finish"], r["zero_unset"]) == (capi.PARTIAL_AVERAGE, 2, f64, capi.FINISH_ROUND, False)
    r = recipe("int32", 0.0, "sum")  # integer sums are exact modulo 2^32 on every path: the typed partials travel
    assert (r["local_method"], r["n_payloads"], r["payload_dtype"][0], r["finish"]) == (capi.METHODS["sum"], 1, capi.DTYPES["int32"], capi.FINISH_NONE)
    r = recipe("int32", nan, "sum")  # the mask is primary: masks are OR-ed (MAX), not added
    assert (r["n_payloads"], r["payload_dtype"][0], r["payload_op"], r["finish"], r["zero_unset"]) == \
        (2, capi.DTYPES["int32"], [capi.XCHG_SUM, capi.XCHG_MAX], capi.FINISH_RESTORE, False)
    for dt in ("float32", "float64", "int32", "uint32"):
        r = recipe(dt, nan, "average")
        assert (r["local_method"], r["payload_dtype"][0], r["payload_op"], r["finish"], r["zero_unset"]) == \
            (capi.PARTIAL_AVERAGE, f64, [capi.XCHG_SUM, capi.XCHG_SUM], capi.FINISH_AVERAGE, False)
    assert f32 != f64
    for m in ("highest", "lowest", "first", "last", "product"):
        assert recipe("float32", nan, m)["n_payloads"] == 1 and recipe("int32", nan, m)["n_payloads"] == 2
        assert recipe("float32", 0.0, m)["finish"] == capi.FINISH_COMBINE
    with pytest.raises(capi.OlapError, match="Unsupported aggregation method"):
        recipe("float32", 0.0, 7)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_drillup_gloo_cpu(tmp_path, world):
    """7 rows over 2 ranks (4 + 3) and 3 ranks (3 + 2 + 2); 2 rows over 3 ranks (one rank has none)."""
    results = run_workers("oracle", tmp_path, world)
    check_dim0(results)
    from olap_in_memory_amd.sharded import partition_rows
    assert results[0]["dice_bounds"] == [int(np.searchsorted([1, 2, 4, 6], b)) for b in partition_rows(7, world)]


def gathered(results, key, n):
    vals, stat = np.full(n, np.nan), np.full(n, -1, np.int64)
    for res in results:
        r = res[key]
        vals[r["range"][0]:r["range"][1]] = r["values"]
        stat[r["range"][0]:r["range"][1]] = r["status"]
    return vals, stat


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_store_shared_gpu(tmp_path, world):
    """The product's sharded store and drillUp op, `world` processes sharing the one GPU, gloo carrying the payloads."""
    results = run_workers("hip", tmp_path, world)
    check_dim0(results)
    case = CASES["f32_zero"]
    for method in ("sum", "average"):  # the whole-result placement agrees with the scattered one, bit for bit
        ev, _ = expected_dim0(case, method)
        assert np.array_equal(np.asarray(results[0]["f32_zero/%s/all" % method]["values"]), ev.astype(np.float64))
    lens = case["lens"]
    o = OracleStore(420, "float32", 0.0)
    o.set_data(case_data(case))
    ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
    sel = lambda l: np.arange(l, dtype=np.int32)  # noqa: E731

    def same(key, store):
        ev, es = expected_typed(store)
        gv, gs = gathered(results, key, ev.size)
        assert np.array_equal(gv, ev.astype(np.float64), equal_nan=True) and np.array_equal(gs, es), key

    same("axis2", o.drill_up(lens, [7, 6, 2], [ident(7), ident(6), (np.arange(10) % 2).astype(np.uint32)], "sum"))
    same("dice12", o.dice(lens, [7, 4, 3], [sel(7), np.array([4, 0, -1, 2], np.int32), np.array([9, 8, 1], np.int32)]))
    same("down2", o.drill_down(lens, [7, 6, 30], [ident(7), ident(6), np.repeat(np.arange(10), 3).astype(np.uint32)], "sum"))
    same("swap12", o.reorder(lens, [0, 2, 1]))
    picked = o.dice(lens, [4, 6, 10], [np.array([1, 2, 4, 6], np.int32), sel(6), sel(10)])
    same("rows", picked)
    same("rows_then_sum", picked.drill_up([4, 6, 10], [4, 1, 10], [ident(4), np.zeros(6, np.uint32), ident(10)], "sum"))
    from olap_in_memory_amd.sharded import partition_rows
    assert results[0]["rows"]["bounds"] == [int(np.searchsorted([1, 2, 4, 6], b)) for b in partition_rows(7, world)]


@pytest.mark.gpu
def test_sharded_single_process():
    """One process: the direct transport (ranks sharing cuda:0, e.g. devices [0, 0, 0]) through
    olap_shard_drillup_step with its event ordering, and RCCL itself on one-rank communicators made by
    olap_comm_init_all and olap_comm_init_rank."""
    r = subprocess.run([sys.executable, os.path.join(HERE, "_sharded_single_process.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "sharded single-process ok" in r.stdout, r.stdout[-6000:]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,world", [([10] * 9, 3), ([320, 5, 5, 5, 5, 5, 5, 10, 20], 4)], ids=["literal-3-ranks", "friendly-4-ranks"])
def test_config4_sharded_on_one_gpu(shape, world):
    """BASELINE configs[3] at full size (10^9 cells), sharded on dim0 over `world` processes that share the one GPU:
    rows 4 + 3 + 3 of the literal [10]^9 (400 MB partials), 80 rows per rank of the shard-friendly shape.  The
    drillUp that collapses the sharded axis runs through the product's sharded store and step; gloo stands in for
    RCCL.  Every rank checks slices of its block of the scattered result against float64 column sums."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29571 + world), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_big_worker.py"), ",".join(str(x) for x in shape)], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    assert "sharded 10^9 ok" in logs[0]


@pytest.mark.gpu
def test_comm_from_torch_process_group():
    """bench.py's way to its RCCL communicator at N > 1 (Comm.from_process_group over a torch "nccl" group), one rank."""
    r = subprocess.run([sys.executable, os.path.join(HERE, "_comm_from_group.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "comm from group ok" in r.stdout, r.stdout[-4000:]
