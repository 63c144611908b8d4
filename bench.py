#!/usr/bin/env python3
"""bench.py — cells aggregated / s on drillUp(sum), MI355X.

  python bench.py [--gpus N --steps K --warmup W]

With N > 1 and no WORLD_SIZE in the environment the script starts its N ranks itself (a child
`python -m torch.distributed.run`, before anything here touches the GPU) and relays rank 0's line;
under torch.distributed.run it is one of the ranks.

A step is one drillUp(sum) of the outermost dimension over a device-resident Float32 measure
(in-memory.js:265-334 through libolapgpu's C ABI):

  N = 1   BASELINE.json's 10^8-cell cube, shape [10]*8, dim0 -> 'all' (the configuration the >= 70 %
          HBM-read target is quoted on).  Reported beside it: configs[1]'s 10^6-cell cube
          (`cache_resident_1e6`), configs[2]'s chain, configs[4]'s calendar roll-up, and configs[3]'s
          WHOLE 10^9-cell cube on this one GPU in both shapes (`config4_single_gpu`) — the N = 1 point
          of the 10^9 scaling series.
  N > 1   configs[3]: the 10^9-cell cube sharded on dim0 (olap_sharded_store / olap_shard_drillup):
          each rank reduces its rows, ONE RCCL reduce-scatter over xGMI combines the partials.
          STRONG scaling: the same 10^9 cells at every N.  Headline shape [320,5,5,5,5,5,5,10,20]
          (40 rows per GPU at N = 8, 12.5 MB partials); the literal [10]*9 (400 MB partials, xGMI-bound
          by construction) is reported beside it as `literal_shape`.

One JSON line on rank 0.  `value` = input cells of the whole job per second, buffers resident in HBM.
`roofline` = algorithmic bytes (4 B read per input cell + 4 B written per output cell) / the kernel's
mean duration from HIP events on the launch stream, against the 8 TB/s peak and against the box's
plain-read ceiling measured in the same run.  `cpu_baseline*` = CPU forms of the same loop timed on
this host on bounded samples; reported baselines, not targets.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
FRIENDLY_SHAPE = [320, 5, 5, 5, 5, 5, 5, 10, 20]  # 10^9 cells; 320 rows divide by 2, 4 and 8
LITERAL_SHAPE = [10] * 9


def cpu_baseline(seconds_budget=12.0):
    """Single-thread C port of the reference's Map-semantics drillUp loop (in-memory.js:265-334: decode D digits, map,
    re-encode, Map get / set per cell) on THE headline configuration itself: [10]*8, 10^8 cells, dimension0 -> all."""
    from oracle.oracle import OracleStore

    lens = [10] * 8
    n = int(np.prod(lens))
    s = OracleStore(n, "float32", 0.0)
    s.fill_seeded(20240807, 1.0)
    maps = [np.zeros(10, np.uint32)] + [np.arange(10, dtype=np.uint32) for _ in lens[1:]]
    reps, spent = 0, 0.0
    while reps < 2 or (spent < seconds_budget and reps < 20):
        t0 = time.perf_counter()
        out = s.drill_up(lens, [1] + lens[1:], maps, "sum")
        spent += time.perf_counter() - t0
        reps += 1
        del out
    return {"value": n * reps / spent, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "the headline workload itself: drillUp(sum) dimension0->all of the [10]*8 float32 cube (1e8 cells, sample ratio 1), "
                      "%d repetitions, oracle/olap_oracle.c (Map-semantics C port of in-memory.js:265-334)" % reps}


def cpu_baseline_flat(seconds_budget=6.0):
    """The flat-TypedArray form (README.md:12-14) as plain C loops on the headline shape [10, 10^7]:
    one thread, and OpenMP over every host core (oracle/flat_baseline.c)."""
    from oracle.oracle import flat_baseline

    L, max_threads = flat_baseline()
    K, inner = 10, 10_000_000
    rng = np.random.default_rng(20240807)
    a = (0.5 + rng.random(K * inner, dtype=np.float32)).astype(np.float32)
    o = np.zeros(inner, np.float32)
    out = {}
    for label, threads in (("single_thread", 1), ("all_cores", max_threads)):
        L.flat_drillup_sum(a.ctypes.data, o.ctypes.data, K, inner, threads)  # warm-up: page faults, thread start
        reps, spent = 0, 0.0
        while reps < 3 or (spent < seconds_budget / 2 and reps < 50):
            spent += L.flat_drillup_sum(a.ctypes.data, o.ctypes.data, K, inner, threads)
            reps += 1
        out[label] = {"value": K * inner * reps / spent, "unit": "cells/s", "cores": threads, "repetitions": reps}
    out["kind"] = "port"
    out["host_cores"] = os.cpu_count()
    out["sample"] = ("drillUp(sum) dim0 of a dense [10, 10000000] float32 buffer (1e8 cells), float64 accumulators, "
                     "oracle/flat_baseline.c (plain loops; OpenMP over column blocks for all_cores)")
    return out


def cpu_baseline_js():
    """The same flat loop in JavaScript over a Float32Array, one thread under node."""
    node = shutil.which("node")
    if not node:
        return {"skipped": "node is not installed on this host"}
    try:
        r = subprocess.run([node, os.path.join(ROOT, "oracle", "flat_baseline.js"), "10", "10000000", "8"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=240)
        if r.returncode != 0:
            return {"skipped": "node failed: " + r.stderr[-300:]}
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as err:  # a baseline problem must not cost the headline line
        return {"skipped": str(err)[:300]}
    return {"value": d["cells_per_s"], "unit": "cells/s", "cores": 1, "host_cores": d["host_cores"], "kind": "port", "node": d["node"],
            "sample": "drillUp(sum) dim0 of a dense Float32Array [10, 10000000] (1e8 cells), %d repetitions, mean; "
                      "oracle/flat_baseline.js (timing shape of test/cube-benchmark.js:5-18)" % d["repetitions"]}


def read_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic_r*.json)."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        import re

        for f in sorted(os.listdir(pdir)):
            if re.fullmatch(r"traffic_r\d+\.json", f):  # the per-round summary of the headline launch (tools/summarize_profiles.py)
                best = os.path.join(pdir, f)
    if not best:
        return None
    try:
        with open(best) as fh:
            return json.load(fh).get("hbm_bytes_per_launch"), os.path.relpath(best, ROOT)
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
    src = store.data_ptr()

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
    # bytes the four rules have to move: sum / average read every day; first / last need the first / last SET day of a
    # month only — on this dense cube one day of ~30 — and the row kernel stops there (one row read per group)
    out1 = 120 * 27400
    b1 = (2 * (n + out1) + 2 * (out1 + out1)) * 4
    per_rule = {}
    for (v, p1, _p2, o1, _o2), method in zip(measures, ("sum", "average", "first", "last")):
        per_rule[method] = _time(torch, lambda: p1.run(v.data_ptr(), None, o1.data_ptr(), None, st), iters=20)
    res = {"cells": n, "measures": 4, "drillUp_time_month_us": us1, "drillUp_time_month_GBps": b1 / (us1 * 1e-6) / 1e9,
           "drillUp_time_month_frac": b1 / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, "cell_measures_per_s": 4 * n / (us1 * 1e-6),
           "drillUp_time_month_algorithmic_bytes": b1, "drillUp_time_month_us_by_rule": {k: round(v, 2) for k, v in per_rule.items()},
           "bytes_note": "sum / average read every cell (4 B) and write the result; first / last read ONE day per month on a dense cube "
                         "(the first / last set member) and write the result",
           "drillUp_location_country_us": us2, "kernel": measures[0][1].kernel_name}
    # the four measures, each with its own rule, as ONE launch (olap_plan_run_batch_rules: what Cube.drillUp sends)
    try:
        rules = ("sum", "average", "first", "last")
        p1, p2 = measures[0][1], measures[0][2]

        def months_one_launch():
            p1.run_batch_rules(rules, [m[0].data_ptr() for m in measures], None, [m[3].data_ptr() for m in measures], None, st)

        def countries_one_launch():
            p2.run_batch_rules(rules, [m[3].data_ptr() for m in measures], None, [m[4].data_ptr() for m in measures], None, st)

        u1, u2 = _time(torch, months_one_launch, iters=20), _time(torch, countries_one_launch, iters=50)
        res["one_launch"] = {"drillUp_time_month_us": u1, "drillUp_time_month_frac": b1 / (u1 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "drillUp_location_country_us": u2, "kernel": "drillup_rows_mixed_kernel"}
    except Exception as err:  # must not cost the headline line
        res["one_launch"] = {"error": str(err)[:200]}
    return res


def bench_1e9_single_gpu(pkg, engine, torch, iters=10, shapes=(("friendly_shape", FRIENDLY_SHAPE), ("literal_shape", LITERAL_SHAPE))):
    """BASELINE configs[3]'s whole 10^9-cell cube on ONE GPU (4 GB resident), both shapes: the N = 1 point of
    the strong-scaling series that `bench.py --gpus N` continues."""
    n = 10 ** 9
    vals = engine.empty(n, "float32")
    pkg.capi.check(pkg.lib().olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 20240807, 1.0, engine.stream()))
    res = {}
    for label, shape in shapes:
        n_out = n // shape[0]
        out = engine.empty(n_out, "float32")
        maps = [np.zeros(shape[0], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in shape[1:]]
        plan = pkg.Plan.drillup("float32", 0.0, "sum", shape, [1] + shape[1:], maps)
        us = _time(torch, lambda: plan.run(vals.data_ptr(), None, out.data_ptr(), None, engine.stream()), iters=iters, warm=2)
        gbs = (n + n_out) * 4 / (us * 1e-6) / 1e9
        res[label] = {"shape": shape, "us_per_step": us, "cells_per_s": n / (us * 1e-6), "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
                      "kernel": plan.kernel_name}
        del out
    del vals
    torch.cuda.empty_cache()
    return res


def read_ceiling(pkg, engine, torch, buf, n_bytes):
    """the achievable read ceiling of THIS box — a plain grid-stride 16-byte streaming read of the
    same 400 MB buffer, measured in the same run (olap_diag_read_ceiling)."""
    scratch = engine.empty(2048, "float32")
    L = pkg.lib()
    us = _time(torch, lambda: pkg.capi.check(L.olap_diag_read_ceiling(buf.data_ptr(), n_bytes, scratch.data_ptr(), engine.stream())), iters=50, warm=5)
    return n_bytes / (us * 1e-6) / 1e9


def write_ceiling(pkg, engine, torch, buf, n_bytes):
    """The write-side counterpart (olap_diag_write_ceiling): 16-byte streaming stores over the output buffer, same run."""
    L = pkg.lib()
    us = _time(torch, lambda: pkg.capi.check(L.olap_diag_write_ceiling(buf.data_ptr(), n_bytes, engine.stream())), iters=50, warm=5)
    return n_bytes / (us * 1e-6) / 1e9


def launch_ranks(args):
    """N > 1 without a launcher: start the ranks as a child process tree BEFORE this process touches the GPU
    (never an exec of a process that has initialised HIP) and relay what they print."""
    import socket

    import torch

    have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if have < args.gpus and not args.rehearse:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s): refusing to run fewer ranks than asked for\n" % (args.gpus, have))
        raise SystemExit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: only the headline workload")
    ap.add_argument("--serial-steps", action="store_true", help="N > 1: no overlap between consecutive steps")
    ap.add_argument("--force-sharded", action="store_true",
                    help="developer aid: run the N > 1 code path (sharded store, RCCL communicator, reduce-scatter) with the ranks given, "
                         "even one — on a 1-GPU box this drives every line of that path through a one-rank RCCL communicator")
    ap.add_argument("--rehearse", action="store_true",
                    help="developer aid: the N ranks share cuda:0 and gloo carries the payloads (exercises the N > 1 code path on "
                         "a 1-GPU box with a 10x smaller cube; the numbers are meaningless)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)

    # RCCL prints a version banner on standard output when a communicator is created; the contract is ONE JSON line
    # there.  With more than one rank (or --force-sharded) everything else this process and its libraries print goes to
    # standard error, and the line is written to the real standard output at the end.
    real_stdout = None
    if args.gpus > 1 or args.force_sharded:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    pkg = load_package()
    from olap_in_memory_amd import capi
    from olap_in_memory_amd.sharded import Comm, HipEngine, ShardedStore, exchange_over_process_group

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libolapgpu has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    capi.check(pkg.lib().olap_set_device(local_rank))
    if world > 1:
        if args.rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    engine = HipEngine(torch.device("cuda", local_rank))
    stream = engine.stream()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    extra = {}
    sharded_path = world > 1 or args.force_sharded
    if not sharded_path:
        lens = [10] * 8
        workload = "drillUp(sum) dimension0->all, 8-dim 10^8-cell Float32 cube [10]*8, all cells set"
        n = int(np.prod(lens))
        n_out = n // lens[0]
        values = engine.empty(n, "float32")
        capi.check(pkg.lib().olap_fill_seeded(values.data_ptr(), None, n, 0, 2, 20240807, 1.0, stream))
        partial = engine.empty(n_out, "float32")
        maps = [np.zeros(lens[0], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in lens[1:]]
        plan = pkg.Plan.drillup("float32", 0.0, "sum", lens, [1] + lens[1:], maps)
        kernel_name = plan.kernel_name
        local_cells = n

        def step():
            plan.run(values.data_ptr(), None, partial.data_ptr(), None, stream)

        def kernel_only():
            step()

        def finish():
            pass
        collective, pipelined, transport = "none", False, "none"
    else:
        lens = list(FRIENDLY_SHAPE)
        fallback_group = None
        if args.rehearse:
            lens[-1] = 2  # 10^8 cells: gloo moves the payloads through host memory
        if args.rehearse:
            comm = Comm.detached(world, rank, 0)
        elif world > 1:
            try:
                comm = Comm.from_process_group(dist, local_rank)
            except pkg.OlapError as err:
                # RCCL could not be bound or initialised by libolapgpu: keep the run alive with a clearly labelled,
                # slow transport (payloads staged through host memory over a gloo group) rather than no record at all
                sys.stderr.write("bench.py: direct RCCL unavailable (%s); falling back to gloo through host memory\n" % err)
                fallback_group = dist.new_group(backend="gloo")
                comm = Comm.detached(world, rank, local_rank)
        else:
            comm = Comm.init_rank(Comm.unique_id(), 1, 0, local_rank)
        transport = comm.transport
        # self-check: the communicator must span exactly the ranks asked for, each on a device of its own
        if comm.world != args.gpus or (world > 1 and dist.get_world_size() != args.gpus):
            sys.stderr.write("bench.py: --gpus %d but the communicator reports %d rank(s) (torch.distributed: %d)\n"
                             % (args.gpus, comm.world, dist.get_world_size() if world > 1 else 1))
            raise SystemExit(3)
        if world > 1 and not args.rehearse:
            mine = torch.tensor([local_rank], dtype=torch.int32, device="cuda")
            devs = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(devs, mine)
            if len({int(d.item()) for d in devs}) != world:
                sys.stderr.write("bench.py: --gpus %d but the ranks sit on %d distinct device(s)\n" % (args.gpus, len({int(d.item()) for d in devs})))
                raise SystemExit(3)
        store = ShardedStore(comm, lens, "float32", 0.0).fill_seeded(20240807, 1.0)
        pipelined = not (args.serial_steps or comm.transport == "detached")
        op = store.plan_drillup_dim0(np.zeros(lens[0], np.uint32), 1, "sum", placement=capi.PLACE_SCATTER, depth=2 if pipelined else 1)
        vals, stat = store.step_inputs()
        n_out = op.out_cells
        local_cells = op.local_cells(0)
        kernel_name = op.kernel_name(0)
        workload = ("drillUp(sum) of the sharded dimension0->all, 9-dim 10^9-cell Float32 cube %s on %d GPUs (%.4g cells per GPU), "
                    "one RCCL reduce-scatter of the partials per step" % (lens, world, local_cells))
        collective = "reduce_scatter"

        if comm.transport == "detached":
            def step():
                op.local(0, vals[0], None, stream)
                exchange_over_process_group(op, dist, fallback_group)
                op.finish(0, stream)
        else:
            def step():
                op.step(vals, stat, [stream])

        def kernel_only():
            op.local(0, vals[0], None, stream)

        def finish():
            op.wait([stream])
    torch.cuda.synchronize()

    # ---- the timed region: W warm-up steps, then exactly K steps between barriers, max over ranks
    for _ in range(args.warmup):
        step()
    finish()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # kernel-only duration on the launch stream (HIP events), separately from the step loop
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    k0.record()
    for _ in range(args.steps):
        kernel_only()
    k1.record()
    torch.cuda.synchronize()
    kernel_ms = k0.elapsed_time(k1) / args.steps

    total_cells = float(np.prod(lens))
    # bytes = N_in*4 (read) + N_out*4 (write); the Int32 mask is neither read nor written here
    # because for Float32 cells over a 0 default it is a function of the values
    alg_bytes = local_cells * 4 + n_out * 4
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    if sharded_path and transport == "rccl" and pipelined:
        # the same steps strictly one after the other (no overlap between the exchange of one query and the local
        # reduction of the next): what a single query costs end to end
        sop = store.plan_drillup_dim0(np.zeros(lens[0], np.uint32), 1, "sum", placement=capi.PLACE_SCATTER, depth=1)
        k_ser = max(2, min(args.steps, 50))
        for _ in range(3):
            sop.step(vals, stat, [stream])
        barrier()
        ts = time.perf_counter()
        for _ in range(k_ser):
            sop.step(vals, stat, [stream])
        barrier()
        ser_elapsed = max_over_ranks(time.perf_counter() - ts)
        extra["serial_steps"] = {"steps": k_ser, "ms_per_step": ser_elapsed / k_ser * 1e3, "cells_per_s": total_cells * k_ser / ser_elapsed,
                                 "note": "depth 1: local reduction, reduce-scatter and finish of each step before the next one starts"}
        del sop
    if sharded_path and transport == "rccl":
        # the literal [10]^9 shape of configs[3] beside the headline: rows split 2,2,1,1,... at N = 8 and every
        # rank ships a 400 MB partial, so the collective dominates; fewer steps, same protocol
        del op, store
        torch.cuda.empty_cache()
        lit = ShardedStore(comm, LITERAL_SHAPE, "float32", 0.0).fill_seeded(20240807, 1.0)
        lop = lit.plan_drillup_dim0(np.zeros(10, np.uint32), 1, "sum", placement=capi.PLACE_SCATTER, depth=2 if pipelined else 1)
        lv, ls = lit.step_inputs()
        k_lit = max(2, min(args.steps, 20))
        for _ in range(2):
            lop.step(lv, ls, [stream])
        lop.wait([stream])
        barrier()
        t1 = time.perf_counter()
        for _ in range(k_lit):
            lop.step(lv, ls, [stream])
        lop.wait([stream])
        barrier()
        lit_elapsed = max_over_ranks(time.perf_counter() - t1)
        extra["literal_shape"] = {"shape": LITERAL_SHAPE, "steps": k_lit, "ms_per_step": lit_elapsed / k_lit * 1e3,
                                  "cells_per_s": 1e9 * k_lit / lit_elapsed, "rows_per_rank": [b - a for a, b in zip(lit.bounds, lit.bounds[1:])],
                                  "partial_bytes_per_rank": int(lop.out_cells) * 8, "partial_type": "float64"}
        del lop, lit

    kernel_ms_per_rank = None
    if sharded_path:
        # every rank's local reduction (HIP events on its own launch stream)
        if world > 1:
            t = torch.tensor([kernel_ms], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            kernel_ms_per_rank = [float(x.item()) for x in gathered]
        else:
            kernel_ms_per_rank = [kernel_ms]
        if transport == "rccl" and not args.rehearse:
            # the N = 1 point of THIS series, measured in THIS run: the whole 10^9-cell cube of the headline shape on rank
            # 0's GPU with the plain one-device plan (the other ranks wait) — so that a reader of the N-GPU line alone, or a
            # sweep whose N = 1 line ran the 10^8 headline, divides like by like
            barrier()
            if rank == 0:
                try:
                    base = bench_1e9_single_gpu(pkg, engine, torch, iters=5, shapes=(("friendly_shape", FRIENDLY_SHAPE),))["friendly_shape"]
                    extra["scaling_base"] = {"n_gpus": 1, "shape": FRIENDLY_SHAPE, "cells": 10 ** 9, "us_per_step": base["us_per_step"],
                                             "cells_per_s": base["cells_per_s"], "kernel": base["kernel"],
                                             "note": "same 10^9 cells on ONE GPU (rank 0's, this run, plain plan): divide `value` by this cells_per_s "
                                                     "for the speed-up; the N = 1 bench line runs the 10^8-cell headline instead"}
                except Exception as err:  # (an out-of-memory here must not cost the line)
                    extra["scaling_base"] = {"error": str(err)[:200]}
            barrier()

    with_mask = None
    ceiling = None
    w_ceiling = None
    if not sharded_path:
        ceiling = read_ceiling(pkg, engine, torch, values, n * 4)
        w_ceiling = write_ceiling(pkg, engine, torch, partial, n_out * 4)  # (the output buffer: rewritten by every step anyway)
        if not args.no_extras:
            # the same launch with the Int32 status mask read and written (10 % of the cells unset)
            sv, ss = engine.empty(n, "float32"), engine.empty(n, "int32")
            capi.check(pkg.lib().olap_fill_seeded(sv.data_ptr(), ss.data_ptr(), n, 0, 2, 20240807, 0.9, stream))
            ost = engine.empty(n_out, "int32")
            mask_us = _time(torch, lambda: plan.run(sv.data_ptr(), ss.data_ptr(), partial.data_ptr(), ost.data_ptr(), stream), iters=args.steps, warm=5)
            mask_bytes = (n + n_out) * 8
            with_mask = {"kernel_ms": mask_us * 1e-3, "algorithmic_bytes": mask_bytes, "achieved_GBps": mask_bytes / (mask_us * 1e-6) / 1e9,
                         "frac": mask_bytes / (mask_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "cells_per_s": n / (mask_us * 1e-6),
                         "note": "values + Int32 status read and written, 90 % of the cells set"}
            del sv, ss, ost

            # configs[1]: 10^6 cells, drillUp on axes 0 / 3 / 5 (resident in the 256 MiB Infinity Cache)
            small = engine.empty(10 ** 6, "float32")
            capi.check(pkg.lib().olap_fill_seeded(small.data_ptr(), None, 10 ** 6, 0, 2, 20240807, 1.0, stream))
            res = {}
            for axis in (0, 3, 5):
                lens6 = [10] * 6
                new6 = list(lens6)
                new6[axis] = 1
                maps6 = [np.zeros(10, np.uint32) if i == axis else np.arange(10, dtype=np.uint32) for i in range(6)]
                o = pkg.Plan.drillup("float32", 0.0, "sum", lens6, new6, maps6)
                ov, os_ = engine.empty(10 ** 5, "float32"), engine.empty(10 ** 5, "int32")
                us = _time(torch, lambda: o.run(small.data_ptr(), None, ov.data_ptr(), os_.data_ptr(), stream), iters=200, warm=20)
                res["axis%d" % axis] = {"us_per_launch": round(us, 3), "cells_per_s": 1e6 / (us * 1e-6)}
                # the same 200 launches captured once into a hipGraph and replayed: what is left is the
                # kernel itself plus the graph's per-node dispatch, without one host launch per query
                try:
                    side = torch.cuda.Stream()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(side):
                        o.run(small.data_ptr(), None, ov.data_ptr(), os_.data_ptr(), side.cuda_stream)
                        side.synchronize()
                        with torch.cuda.graph(graph, stream=side):
                            for _ in range(200):
                                o.run(small.data_ptr(), None, ov.data_ptr(), os_.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    graph.replay()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
            # the same headline roll-up over FLOAT64 cells — what Float64 measures are, and what the Node host keeps int32 /
            # uint32 measures in (the reference's Map holds float64 numbers whatever the declared type): 880 MB per launch
            try:
                v64 = engine.empty(n, "float64")
                capi.check(pkg.lib().olap_fill_seeded(v64.data_ptr(), None, n, 0, 3, 20240807, 1.0, stream))
                o64 = engine.empty(n_out, "float64")
                p64 = pkg.Plan.drillup("float64", 0.0, "sum", lens, [1] + lens[1:], maps)
                us64 = _time(torch, lambda: p64.run(v64.data_ptr(), None, o64.data_ptr(), None, stream), iters=args.steps, warm=5)
                b64 = (n + n_out) * 8
                extra["float64_cells"] = {"kernel_ms": us64 * 1e-3, "algorithmic_bytes": b64, "achieved_GBps": b64 / (us64 * 1e-6) / 1e9,
                                          "frac": b64 / (us64 * 1e-6) / 1e9 / HBM_PEAK_GBS, "cells_per_s": n / (us64 * 1e-6), "kernel": p64.kernel_name,
                                          "note": "the headline shape with 8-byte cells (Float64 measures; integer measures of the Node host)"}
                del v64, o64, p64
                torch.cuda.empty_cache()
            except Exception as err:  # (must not cost the headline line)
                extra["float64_cells"] = {"error": str(err)[:200]}
            # several measures of a cube that share a rule: one launch for all (olap_plan_run_batch) against one
            # launch per measure, on the 10^6-cell cube (launch-bound) — what Cube.drillUp does per stored measure
            try:
                nm = 4
                new6 = [1] + lens6[1:]
                maps6 = [np.zeros(10, np.uint32)] + [np.arange(10, dtype=np.uint32) for _ in range(5)]
                o = pkg.Plan.drillup("float32", 0.0, "sum", lens6, new6, maps6)
                ins = [engine.empty(10 ** 6, "float32") for _ in range(nm)]
                for t in ins:
                    capi.check(pkg.lib().olap_fill_seeded(t.data_ptr(), None, 10 ** 6, 0, 2, 20240807, 1.0, stream))
                outs = [engine.empty(10 ** 5, "float32") for _ in range(nm)]
                ip, op = [t.data_ptr() for t in ins], [t.data_ptr() for t in outs]

                def one_by_one():
                    for i in range(nm):
                        o.run(ip[i], None, op[i], None, stream)

                us1 = _time(torch, one_by_one, iters=200, warm=20)
                usb = _time(torch, lambda: o.run_batch(ip, None, op, None, stream), iters=200, warm=20)
                rules4 = ("sum", "average", "first", "last")
                plans4 = [pkg.Plan.drillup("float32", 0.0, r, lens6, new6, maps6) for r in rules4]

                def rule_by_rule():
                    for i in range(nm):
                        plans4[i].run(ip[i], None, op[i], None, stream)

                usr1 = _time(torch, rule_by_rule, iters=200, warm=20)
                usr = _time(torch, lambda: o.run_batch_rules(rules4, ip, None, op, None, stream), iters=200, warm=20)
                # the same four rules rolled up along an INNER dimension (axis 3: the row-tile regime)
                new6i = lens6[:3] + [1] + lens6[4:]
                maps6i = [np.arange(10, dtype=np.uint32) for _ in range(3)] + [np.zeros(10, np.uint32)] + [np.arange(10, dtype=np.uint32) for _ in range(2)]
                plans4i = [pkg.Plan.drillup("float32", 0.0, r, lens6, new6i, maps6i) for r in rules4]

                def rule_by_rule_inner():
                    for i in range(nm):
                        plans4i[i].run(ip[i], None, op[i], None, stream)

                usi1 = _time(torch, rule_by_rule_inner, iters=200, warm=20)
                usi = _time(torch, lambda: plans4i[0].run_batch_rules(rules4, ip, None, op, None, stream), iters=200, warm=20)
                extra["batched_measures_1e6"] = {"measures": nm, "one_launch_each_us": round(us1, 3), "one_launch_for_all_us": round(usb, 3),
                                                 "four_rules_inner_axis": {"axis": 3, "one_launch_each_us": round(usi1, 3), "one_launch_for_all_us": round(usi, 3),
                                                                           "kernel": "drillup_tile_mixed_kernel"},
                                                 "cell_measures_per_s": nm * 1e6 / (usb * 1e-6), "kernel": o.kernel_name,
                                                 "four_rules": {"rules": list(rules4), "one_launch_each_us": round(usr1, 3),
                                                                "one_launch_for_all_us": round(usr, 3), "kernel": "drillup_rows_mixed_kernel"}}
                del ins, outs
            except Exception as err:  # must not cost the headline line
                extra["batched_measures_1e6"] = {"error": str(err)[:200]}
            extra["config3_chain"] = bench_config3(pkg, engine, values, torch)
            extra["config5_time_rollup"] = bench_config5(pkg, engine, torch)
            del values, partial
            torch.cuda.empty_cache()
            extra["config4_single_gpu"] = bench_1e9_single_gpu(pkg, engine, torch)

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
            # N = 1 is the 10^8-cell headline; every N > 1 runs the SAME 10^9 cells (config4_single_gpu is their N = 1 point)
            "scaling": "strong" if sharded_path else "weak",
            "vs_baseline": None,
            "dtype": "f64",  # the arithmetic type: float64 accumulators over Float32 cells (config.cell_type)
            "data": "synthetic (seeded mulberry32, values in [0.5,1.5), generated on device)",
            "config": {"workload": workload, "shape": lens, "cells_per_gpu": local_cells, "cell_type": "float32",
                       "kernel": kernel_name, "collective": collective, "transport": transport, "steps_pipelined": bool(pipelined)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes,
                         "hbm_read_frac": local_cells * 4 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if not sharded_path:
            traffic = read_traffic()
            if traffic:
                line["roofline"]["traffic"] = traffic[0]
                line["roofline"]["traffic_source"] = ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (gfx950 x2 FETCH_SIZE correction), "
                                                      "recorded when the profile was taken — NOT measured in this run" % traffic[1])
        else:
            # what travels: every rank ships its float64 ACCUMULATORS (the reference never rounds between contributions)
            line["config"].update({"partial_type": "float64", "partial_bytes_per_rank": int(n_out) * 8,
                                   "result_cells_per_rank": int(n_out) // world if world else int(n_out)})
            line["roofline"]["kernel_ms_per_rank"] = kernel_ms_per_rank
            line["cpu_baseline_ref"] = ("timed on rank 0 at N = 1 only (the contract): see `cpu_baseline` of the N = 1 line of this sweep "
                                        "(BENCH_rNN.json); a reported baseline, not a target")
        if ceiling:
            line["roofline"].update({"read_ceiling": ceiling, "frac_of_read_ceiling": achieved / ceiling,
                                     "read_ceiling_note": "plain 16-byte streaming read of the same 400 MB buffer, same run"})
        if ceiling and w_ceiling:
            # HBM is half-duplex: R bytes read and W bytes written cost R / read ceiling + W / write ceiling on this box
            # (tools/headline_limit.hip, profiles/headline_ab_r03.txt), which is what bounds this roll-up — not (R + W) / peak
            bound_us = (n * 4 / ceiling + n_out * 4 / w_ceiling) * 1e-3
            line["roofline"].update({"write_ceiling": w_ceiling,
                                     "half_duplex_bound": {"us": bound_us, "frac_of_peak": alg_bytes / (bound_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                           "achieved_over_bound": bound_us / (kernel_ms * 1e3),
                                                           "note": "read time at the box's read ceiling + write time at its write ceiling, both measured in this run"}})
        if with_mask:
            line["with_status_mask"] = with_mask
        line.update(extra)
        if not sharded_path and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["cpu_baseline_flat"] = cpu_baseline_flat()
            line["cpu_baseline_js"] = cpu_baseline_js()
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
