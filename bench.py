#!/usr/bin/env python3
"""bench.py — cells aggregated / s on drillUp(sum), MI355X.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

A step is one drillUp(sum) of the outermost dimension over a device-resident Float32 measure
(in-memory.js:265-334 through libolapgpu's C ABI):

  N = 1   BASELINE.json's 10^8-cell cube, shape [10]*8, dim0 -> 'all'   (the configuration the
          >= 70 % HBM-read target is quoted on; configs[1]'s 10^6-cell cube fits the Infinity Cache
          and is reported beside it as `cache_resident_1e6`)
  N > 1   the 10^9-cell family sharded on dim0, 40 rows per GPU: [40*N,5,5,5,5,5,5,10,20]
          (1.25e8 cells per GPU, = configs[3]'s shard-friendly shape at N = 8); each rank reduces
          its rows, one RCCL reduce-scatter over xGMI combines the partials.  Weak scaling.

One JSON line on rank 0.  `value` = input cells of all ranks per second, buffers resident in HBM.
`roofline` = algorithmic bytes (4 B read per input cell + 4 B value and 4 B status written per
output cell) / the kernel's mean duration from HIP events on the launch stream.  `cpu_baseline` =
the CPU oracle (oracle/olap_oracle.c, a single-thread C port of the reference loop) timed on this
host on a bounded sample; a reported baseline, not a target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(seconds_budget=12.0):
    """Single-thread C port of the reference's drillUp loop on [10, 2*10^6] (2*10^7 cells)."""
    from oracle.oracle import OracleStore

    lens = [10, 2_000_000]
    n = lens[0] * lens[1]
    s = OracleStore(n, "float32", 0.0)
    s.fill_seeded(20240807, 1.0)
    maps = [np.zeros(10, np.uint32), np.arange(lens[1], dtype=np.uint32)]
    reps, spent = 0, 0.0
    while reps < 2 or (spent < seconds_budget and reps < 20):
        t0 = time.perf_counter()
        out = s.drill_up(lens, [1, lens[1]], maps, "sum")
        spent += time.perf_counter() - t0
        reps += 1
        del out
    return {"value": n * reps / spent, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "drillUp(sum) dim0 of a [10, 2000000] float32 cube (2e7 cells), %d repetitions, "
                      "oracle/olap_oracle.c (Map-semantics C port of in-memory.js:265-334)" % reps}


def read_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic_r*.json)."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.startswith("traffic_r") and f.endswith(".json"):
                best = os.path.join(pdir, f)
    if not best:
        return None
    try:
        with open(best) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _time(torch, fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def bench_config3(pkg, engine, store, torch):
    """BASELINE configs[2]: slice(dimension1,item3) -> dice(dimension4,[1,4,7]) -> drillUp(dimension0,all)
    on the resident 10^8-cell cube, as two fused dice->drillUp launches (K5) and as four plain ones."""
    shape = [10] * 8
    ident = lambda lens: [np.arange(l, dtype=np.int32) for l in lens]
    umap = lambda lens, axis: [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    sel1 = ident(shape)
    sel1[1] = np.array([3], np.int32)
    mid1 = [10, 1, 10, 10, 10, 10, 10, 10]
    sel2 = ident(mid1)
    sel2[4] = np.array([1, 4, 7], np.int32)
    mid2 = list(mid1)
    mid2[4] = 3
    new2 = [1] + mid2[1:]
    P = pkg.Plan
    f1 = P.dice_drillup("float32", 0.0, "sum", shape, mid1, mid1, sel1, umap(mid1, 1))
    f2 = P.dice_drillup("float32", 0.0, "sum", mid1, mid2, new2, sel2, umap(mid2, 0))
    u1, u2 = P.dice("float32", 0.0, shape, mid1, sel1), P.drillup("float32", 0.0, "sum", mid1, mid1, umap(mid1, 1))
    u3, u4 = P.dice("float32", 0.0, mid1, mid2, sel2), P.drillup("float32", 0.0, "sum", mid2, new2, umap(mid2, 0))
    t1, t2 = engine.empty(10 ** 7, "float32"), engine.empty(10 ** 7, "float32")
    t3, t4 = engine.empty(3 * 10 ** 6, "float32"), engine.empty(3 * 10 ** 5, "float32")
    st = engine.stream()
    src = store.values.data_ptr()

    # the whole chain as ONE selection over the source cube (what the Node host issues: the slice's
    # single-member roll-up moves no cells, so its dice stays pending and composes with the next one)
    sel_c = ident(shape)
    sel_c[1] = np.array([3], np.int32)
    sel_c[4] = np.array([1, 4, 7], np.int32)
    c1 = P.dice_drillup("float32", 0.0, "sum", shape, mid2, new2, sel_c, umap(mid2, 0))

    def composed():
        c1.run(src, None, t4.data_ptr(), None, st)

    def fused():
        f1.run(src, None, t1.data_ptr(), None, st)
        f2.run(t1.data_ptr(), None, t4.data_ptr(), None, st)

    def unfused():
        u1.run(src, None, t1.data_ptr(), None, st)
        u2.run(t1.data_ptr(), None, t2.data_ptr(), None, st)
        u3.run(t2.data_ptr(), None, t3.data_ptr(), None, st)
        u4.run(t3.data_ptr(), None, t4.data_ptr(), None, st)

    us_f, us_u = _time(torch, fused), _time(torch, unfused)
    ref4 = t4.clone()
    us_c = _time(torch, composed)
    if not torch.equal(ref4, t4):
        raise SystemExit("config 3: the composed launch disagrees with the four-launch chain")
    composed_bytes = (3 * 10 ** 6 + 3 * 10 ** 5) * 4  # surviving cells read once, result written
    surviving = 3 * 10 ** 6  # cells of the cube that reach the final drillUp
    fused_bytes = (10 ** 7 + 10 ** 7 + 3 * 10 ** 6 + 3 * 10 ** 5) * 4  # read 1e7, write 1e7, read 3e6, write 3e5
    return {"composed_us": us_c, "composed_algorithmic_bytes": composed_bytes, "composed_GBps": composed_bytes / (us_c * 1e-6) / 1e9,
            "fused_us": us_f, "unfused_us": us_u, "launches": {"composed": 1, "fused": 2, "unfused": 4},
            "fused_algorithmic_bytes": fused_bytes, "fused_GBps": fused_bytes / (us_f * 1e-6) / 1e9,
            "surviving_cells": surviving, "full_cube_cells": 10 ** 8,
            "note": "after the slice the working set (40 MB) sits in the 256 MiB Infinity Cache"}


def bench_config5(pkg, engine, torch):
    """BASELINE configs[4]: time(day, 2010-2019)=3652 x location(city)=100 x sku=274 (1.0006e8 cells),
    4 measures with sum / average / first / last; drillUp(time, month) then drillUp(location, country)."""
    import datetime

    lens = [3652, 100, 274]
    n = int(np.prod(lens))
    d0 = datetime.date(2010, 1, 1)
    day_to_month = np.array([(d0 + datetime.timedelta(days=i)).month - 1 + 12 * ((d0 + datetime.timedelta(days=i)).year - 2010) for i in range(3652)], np.uint32)
    city_to_country = (np.arange(100) // 10).astype(np.uint32)
    ident = lambda l: np.arange(l, dtype=np.uint32)
    measures = []
    st = engine.stream()
    for m, method in enumerate(("sum", "average", "first", "last")):
        v = engine.empty(n, "float32")
        pkg.capi.check(pkg.lib().olap_fill_seeded(v.data_ptr(), None, n, 0, 2, 20240807 + m, 1.0, st))
        p1 = pkg.Plan.drillup("float32", 0.0, method, lens, [120, 100, 274], [day_to_month, ident(100), ident(274)])
        p2 = pkg.Plan.drillup("float32", 0.0, method, [120, 100, 274], [120, 10, 274], [ident(120), city_to_country, ident(274)])
        measures.append((v, p1, p2, engine.empty(120 * 27400, "float32"), engine.empty(120 * 2740, "float32")))

    def months():
        for v, p1, _p2, o1, _o2 in measures:
            p1.run(v.data_ptr(), None, o1.data_ptr(), None, st)

    def countries():
        for _v, _p1, p2, o1, o2 in measures:
            p2.run(o1.data_ptr(), None, o2.data_ptr(), None, st)

    us1, us2 = _time(torch, months, iters=20), _time(torch, countries, iters=50)
    b1 = 4 * (n + 120 * 27400) * 4
    return {"cells": n, "measures": 4, "drillUp_time_month_us": us1, "drillUp_time_month_GBps": b1 / (us1 * 1e-6) / 1e9,
            "drillUp_time_month_frac": b1 / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, "cell_measures_per_s": 4 * n / (us1 * 1e-6),
            "drillUp_location_country_us": us2, "kernel": measures[0][1].kernel_name}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-steps", action="store_true", help="N > 1: no overlap between consecutive steps")
    ap.add_argument("--rehearse", action="store_true",
                    help="developer aid: N ranks share cuda:0 and talk over gloo (exercises the N>1 code path on a 1-GPU box; numbers are meaningless)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    pkg = load_package()
    from olap_in_memory_amd.sharded import HipEngine, ShardedStore

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libolapgpu has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    pkg.capi.check(pkg.lib().olap_set_device(local_rank))
    if world > 1:
        if args.rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    engine = HipEngine(torch.device("cuda", local_rank))

    if world == 1:
        lens = [10] * 8
        workload = "drillUp(sum) dimension0->all, 8-dim 10^8-cell Float32 cube [10]*8, all cells set"
    else:
        lens = [40 * world, 5, 5, 5, 5, 5, 5, 10, 20]
        workload = ("drillUp(sum) of the sharded dimension0->all, 9-dim cube [%d,5,5,5,5,5,5,10,20] "
                    "(1.25e8 cells per GPU, %.3g cells), RCCL reduce-scatter of the partials" % (lens[0], float(np.prod(lens))))
    store = ShardedStore(lens, "float32", 0.0, rank, world, engine).fill_seeded(20240807, 1.0)
    op = store.plan_drillup_dim0(np.zeros(lens[0], np.uint32), 1, "sum")
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # At N > 1 the K steps are independent queries over the resident shards; they are issued as a
    # pipeline (the reduce-scatter of step i overlaps the local reduction of step i+1, two buffer
    # pairs).  --serial-steps times them strictly one after the other instead.
    step = op.step if (world == 1 or args.serial_steps) else op.step_pipelined
    for _ in range(args.warmup):
        step()
    op.flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    op.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel-only duration on the launch stream (HIP events), separately from the step loop
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    k0.record()
    for _ in range(args.steps):
        op.local.run(store.values, None, op.partial, None)
    k1.record()
    torch.cuda.synchronize()
    kernel_ms = k0.elapsed_time(k1) / args.steps

    local_cells = store.local_cells
    total_cells = float(np.prod(lens))
    n_out = op.n_out
    # bytes = N_in*4 (read) + N_out*4 (write); the Int32 mask is neither read nor
    # written here because for Float32 cells over a 0 default it is a function of the values
    alg_bytes = local_cells * 4 + n_out * 4
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    # the same launch with the Int32 status mask read and written (10 % of the cells unset)
    with_mask = None
    if world == 1:
        sparse = ShardedStore(lens, "float32", 0.0, rank, world, engine).fill_seeded(20240807, 0.9)
        ost = engine.empty(n_out, "int32")
        for _ in range(5):
            op.local.run(sparse.values, sparse.status, op.partial, ost)
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m0.record()
        for _ in range(args.steps):
            op.local.run(sparse.values, sparse.status, op.partial, ost)
        m1.record()
        torch.cuda.synchronize()
        mask_ms = m0.elapsed_time(m1) / args.steps
        mask_bytes = (local_cells + n_out) * 8
        with_mask = {"kernel_ms": mask_ms, "algorithmic_bytes": mask_bytes, "achieved_GBps": mask_bytes / (mask_ms * 1e-3) / 1e9,
                     "frac": mask_bytes / (mask_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "cells_per_s": local_cells / (mask_ms * 1e-3),
                     "note": "values + Int32 status read and written, 90 % of the cells set"}
        del sparse

    extra = {}
    if world == 1:
        # configs[1]: 10^6 cells, drillUp on axes 0 / 3 / 5 (resident in the 256 MiB Infinity Cache)
        small = ShardedStore([10] * 6, "float32", 0.0, 0, 1, engine).fill_seeded(20240807, 1.0)
        res = {}
        for axis in (0, 3, 5):
            lens6 = [10] * 6
            new6 = list(lens6)
            new6[axis] = 1
            maps = [np.zeros(10, np.uint32) if i == axis else np.arange(10, dtype=np.uint32) for i in range(6)]
            o = engine.make_drillup("float32", 0.0, "sum", lens6, new6, maps)
            ov, os_ = engine.empty(10 ** 5, "float32"), engine.empty(10 ** 5, "int32")
            for _ in range(20):
                o.run(small.values, None, ov, os_)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                o.run(small.values, None, ov, os_)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 200 * 1e3
            res["axis%d" % axis] = {"us_per_launch": round(us, 3), "cells_per_s": 1e6 / (us * 1e-6)}
            # the same 200 launches captured once into a hipGraph and replayed: what is left is the
            # kernel itself plus the graph's per-node dispatch, without one host launch per query
            try:
                side = torch.cuda.Stream()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    o.plan.run(small.values.data_ptr(), None, ov.data_ptr(), os_.data_ptr(), side.cuda_stream)
                    side.synchronize()
                    with torch.cuda.graph(graph, stream=side):
                        for _ in range(200):
                            o.plan.run(small.values.data_ptr(), None, ov.data_ptr(), os_.data_ptr(), torch.cuda.current_stream().cuda_stream)
                graph.replay()
                torch.cuda.synchronize()
                a.record()
                for _ in range(5):
                    graph.replay()
                b.record()
                torch.cuda.synchronize()
                gus = a.elapsed_time(b) / 1000 * 1e3
                res["axis%d" % axis].update({"graph_us_per_launch": round(gus, 3), "graph_cells_per_s": 1e6 / (gus * 1e-6)})
            except Exception as err:  # a capture problem must not cost the headline line
                res["axis%d" % axis]["graph_error"] = str(err)[:200]
        extra["cache_resident_1e6"] = res
        extra["config3_chain"] = bench_config3(pkg, engine, store, torch)
        extra["config5_time_rollup"] = bench_config5(pkg, engine, torch)

    if rank == 0:
        line = {
            "metric": "cells aggregated/sec on drillUp(sum)",
            "value": total_cells * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",  # the arithmetic type: float64 accumulators over Float32 cells (config.cell_type)
            "data": "synthetic (seeded mulberry32, values in [0.5,1.5), generated on device)",
            "config": {"workload": workload, "shape": lens, "cells_per_gpu": local_cells, "cell_type": "float32",
                       "kernel": op.local.plan.kernel_name, "collective": ("reduce_scatter" if op.scatter else "all_reduce") if world > 1 else "none",
                       "steps_pipelined": bool(world > 1 and not args.serial_steps)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": read_traffic() if world == 1 else None,
                         "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes,
                         "hbm_read_frac": local_cells * 4 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if with_mask:
            line["with_status_mask"] = with_mask
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
