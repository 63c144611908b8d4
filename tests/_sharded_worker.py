"""Worker for tests/test_sharded_gloo.py: one rank of a gloo process group.

engine "oracle" (CPU, no GPU needed): the partition (olap_shard_bounds), the row sub-maps and the recipe
(olap_shard_recipe_get: what is shipped, with which reduction, how it is finished) come from
libolapgpu's host-only entry points; the local cell arithmetic is played by the CPU oracle and gloo
carries the payloads.  This rehearses the MATHS of the sharded drillUp for every store the reference
allows (NaN / 0 default, float / integer cells) without a GPU.

engine "hip" (GPU box, the ranks share cuda:0): the product path itself — olap_sharded_store +
olap_shard_drillup on a detached communicator — with gloo standing in for RCCL, which refuses two
ranks on one device."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from oracle.oracle import OracleStore, to_typed  # noqa: E402
from sharded_cases import CASES, case_data, methods_of  # noqa: E402

pkg = load_package()
from olap_in_memory_amd import capi  # noqa: E402
from olap_in_memory_amd import sharded  # noqa: E402

NP = {"int32": np.int32, "uint32": np.uint32, "float32": np.float32, "float64": np.float64}


def is_default(v, dtype, default_nan):
    if dtype in ("int32", "uint32"):
        return np.zeros(v.shape, bool) if default_nan else (v == 0)
    return np.isnan(v) if default_nan else (v == 0)


def default_typed(dtype, default_nan):
    return NP[dtype](np.nan) if (default_nan and dtype.startswith("float")) else NP[dtype](0)


def oracle_of(values_f64, dtype, default):
    o = OracleStore(values_f64.size, dtype, default)
    o.set_data(values_f64)
    return o


def typed_to_f64(tv, ts, dtype, default_nan):
    """typed values + mask -> the float64 `data` an oracle store is filled from (unset = default)."""
    v = tv.astype(np.float64)
    if default_nan:
        v = np.where(ts == 2, v, np.nan)
    return v


def all_reduce(arr, op):
    wire = arr.view(np.int32) if arr.dtype == np.uint32 else arr
    t = torch.from_numpy(np.ascontiguousarray(wire).copy())
    dist.all_reduce(t, op=op)
    return t.numpy().view(arr.dtype)


def all_gather(arr, world):
    wire = arr.view(np.int32) if arr.dtype == np.uint32 else arr
    t = torch.from_numpy(np.ascontiguousarray(wire).copy())
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.concatenate([o.numpy() for o in outs]).view(arr.dtype)


def rehearse_dim0_drillup(full, lens, dtype, default, row_map, n_groups, method, rank, world):
    """The sharded drillUp of dimension 0 as olap_sharded.hip composes it, on the CPU."""
    default_nan = default != default
    rec = sharded.recipe(dtype, default, method)
    bounds = sharded.partition_rows(lens[0], world)
    inner0 = int(np.prod(lens[1:]))
    lo, hi = bounds[rank], bounds[rank + 1]
    new_len = [n_groups] + lens[1:]
    n_out = n_groups * inner0
    maps = [np.asarray(row_map[lo:hi], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in lens[1:]]
    local_lens = [hi - lo] + lens[1:]
    slab = full[lo * inner0:hi * inner0]
    # ---- local partial
    if hi > lo:
        o = oracle_of(slab, dtype, default)
        if rec["local_method"] == capi.PARTIAL_AVERAGE:
            # the float64 ACCUMULATOR of the rank (never rounded to the cell type), 0 where nothing contributed
            cells, present = o.dense()
            acc = oracle_of(np.where(present, cells, default), "float64", default)
            pv, pset = acc.drill_up(local_lens, new_len, maps, "sum").dense()
            tv = np.where(pset, pv, 0.0)
            ones = oracle_of(np.where(present, 1.0, 0.0), "float64", 0.0)
            flag = ones.drill_up(local_lens, new_len, maps, "sum").dense()[0].astype(np.int32)  # contribution counts
        else:
            names = {v: k for k, v in capi.METHODS.items()}
            tv, flag = o.drill_up(local_lens, new_len, maps, names[rec["local_method"]]).typed()
    elif rec["local_method"] == capi.PARTIAL_AVERAGE:
        tv, flag = np.zeros(n_out, np.float64), np.zeros(n_out, np.int32)
    else:
        tv, flag = np.full(n_out, default_typed(dtype, default_nan)), np.zeros(n_out, np.int32)
    assert tv.dtype == NP[sharded.NAME_OF_DTYPE[rec["payload_dtype"][0]]], (tv.dtype, rec)
    if rec["zero_unset"]:
        tv = np.where((flag != 0) & ~np.isnan(tv.astype(np.float64)), tv, NP[dtype](0)).astype(NP[dtype])
    payloads = [tv, flag][: rec["n_payloads"]]
    # ---- exchange
    got = []
    for p, arr in enumerate(payloads):
        op = rec["payload_op"][p]
        if op == capi.XCHG_GATHER:
            got.append(all_gather(arr, world))
        else:
            got.append(all_reduce(arr, dist.ReduceOp.SUM if op == capi.XCHG_SUM else dist.ReduceOp.MAX))
    # ---- finish
    if rec["finish"] == capi.FINISH_NONE:
        v = got[0]
        st = np.where(is_default(v, dtype, default_nan), 0, 2).astype(np.int32)
    elif rec["finish"] == capi.FINISH_RESTORE:
        v, fl = got[0], got[1]
        is_set = ((fl & 2) != 0) & ~is_default(v, dtype, default_nan)
        v = np.where(is_set, v, default_typed(dtype, default_nan)).astype(NP[dtype])
        st = np.where(is_set, 2, 0).astype(np.int32)
    elif rec["finish"] in (capi.FINISH_AVERAGE, capi.FINISH_ROUND):
        # float64 sums added over the ranks (+ contribution counts), rounded to the cell type ONCE
        r = got[0]
        assert r.dtype == np.float64
        counts = got[1] if rec["n_payloads"] > 1 else np.ones(r.size, np.int32)
        d = np.nan if default_nan else 0.0
        def64 = (lambda x: np.isnan(x)) if default_nan else (lambda x: x == 0)
        has = (counts != 0) & ~def64(r)
        if rec["finish"] == capi.FINISH_AVERAGE:
            c16 = counts & 0xFFFF
            with np.errstate(invalid="ignore", divide="ignore"):
                r = np.where(c16 != 0, np.where(has, r, d) / np.maximum(c16, 1), r)
            has = np.where(c16 != 0, ~def64(r), has)
        tvv = to_typed(np.where(has, r, d), dtype)
        is_set = has & ~is_default(tvv, dtype, default_nan)
        v = np.where(is_set, tvv, default_typed(dtype, default_nan)).astype(NP[dtype])
        st = np.where(is_set, 2, 0).astype(np.int32)
    else:  # FINISH_COMBINE: the same drillUp over the rank axis
        gv = got[0]
        gs = got[1] if rec["n_payloads"] > 1 else np.where(is_default(gv, dtype, default_nan), 0, 2)
        o = oracle_of(typed_to_f64(gv, gs, dtype, default_nan), dtype, default)
        v, st = o.drill_up([world, n_out], [1, n_out], [np.zeros(world, np.uint32), np.arange(n_out, dtype=np.uint32)], method).typed()
    return v, st


def hip_dim0_drillup(store, row_map, n_groups, method, placement, depth=1):
    op = store.plan_drillup_dim0(row_map, n_groups, method, placement=placement, depth=depth)
    vals, stat = store.step_inputs()
    op.local(0, vals[0], stat[0] if stat else None)
    sharded.exchange_over_process_group(op, dist)
    op.finish(0)
    v, st, first = op.result_host(0)
    return v, st, first


def dump_sharded(results, key, st):
    lo, hi = st.local_range(0)
    results[key] = {"range": [lo, hi], "values": st.get_data_f64()[lo:hi].tolist(), "status": st.get_status()[lo:hi].tolist(),
                    "bounds": st.bounds, "lens": st.lens}


def main():
    engine, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    results = {}
    comm = None
    if engine == "hip":
        capi.check(capi.lib().olap_set_device(0))
        comm = sharded.Comm.detached(world, rank, 0)
    for name, case in CASES.items():
        lens, dtype, default, row_map, n_groups = case["lens"], case["dtype"], case["default"], case["row_map"], case["groups"]
        full = case_data(case)
        n_out = n_groups * int(np.prod(lens[1:]))
        store = None
        if engine == "hip":
            store = sharded.ShardedStore(comm, lens, dtype, default).set_data_f64(full)
        for method in methods_of(case):
            if engine == "oracle":
                v, st = rehearse_dim0_drillup(full, lens, dtype, default, row_map, n_groups, method, rank, world)
                first = 0
            else:
                # additive methods: the scattered placement bench.py uses; the others arrive whole
                v, st, first = hip_dim0_drillup(store, row_map, n_groups, method, capi.PLACE_SCATTER)
                if st is None:
                    st = np.where(v != 0, 2, 0).astype(np.int32)  # FINISH_NONE: the mask is a function of the values
                if rank == 0 and method in ("sum", "average"):
                    v2, st2, f2 = hip_dim0_drillup(store, row_map, n_groups, method, capi.PLACE_ALL)
                    assert f2 == 0 and v2.size == n_out
                    results["%s/%s/all" % (name, method)] = {"first": 0, "values": v2.astype(np.float64).tolist()}
                elif method in ("sum", "average"):
                    hip_dim0_drillup(store, row_map, n_groups, method, capi.PLACE_ALL)  # collectives are collective
            results["%s/%s" % (name, method)] = {"first": int(first), "values": np.asarray(v, np.float64).tolist(),
                                                 "status": np.asarray(st).tolist()}
        if engine == "hip" and name == "f32_zero":
            ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
            # non-sharded axes: per shard, no communication, the partition is kept
            dump_sharded(results, "axis2", store.drill_up([7, 6, 2], [ident(7), ident(6), (np.arange(10) % 2).astype(np.uint32)], "sum"))
            dump_sharded(results, "dice12", store.dice([7, 4, 3], [np.arange(7), [4, 0, -1, 2], [9, 8, 1]]))
            dump_sharded(results, "down2", store.drill_down([7, 6, 30], [ident(7), ident(6), np.repeat(np.arange(10), 3)], "sum"))
            dump_sharded(results, "swap12", store.reorder([0, 2, 1]))
            picked = store.dice([4, 6, 10], [[1, 2, 4, 6], np.arange(6), np.arange(10)])
            dump_sharded(results, "rows", picked)
            dump_sharded(results, "rows_then_sum", picked.drill_up([4, 1, 10], [ident(4), np.zeros(6, np.uint32), ident(10)], "sum"))
            for bad, args in (("reorder", ([1, 0, 2],)), ("dice", ([2, 6, 10], [[3, 1], np.arange(6), np.arange(10)]))):
                try:
                    getattr(store, bad)(*args)
                    raise AssertionError("expected a 'sharded:' refusal")
                except capi.OlapError as e:
                    assert "sharded:" in str(e), str(e)
    # host-only partition arithmetic, wherever it runs
    results["dice_bounds"] = sharded.dice_bounds(sharded.partition_rows(7, world), [1, 2, 4, 6])
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
