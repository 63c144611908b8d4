#!/usr/bin/env python3
"""Developer tool: the operations of a realistic cube whose extents are all odd —
[3653 days, 101 locations, 271 products], 10^8 Float32 cells — where no row is a whole number of
16-byte groups and every kernel runs its 4-byte-per-lane form."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from olap_in_memory_amd.sharded import HipEngine  # noqa: E402

eng = HipEngine("cuda:0")
L = pkg.lib()
DT = os.environ.get("DTYPE", "float32")  # float32 | float64
ELEM, CODE = {"float32": (4, 2), "float64": (8, 3)}[DT]
shape = [3653, 101, 271]
n = int(np.prod(shape))
days = np.arange(np.datetime64("2010-01-01"), np.datetime64("2010-01-01") + shape[0])
month = days.astype("datetime64[M]").astype(np.int64)
month = (month - month[0]).astype(np.uint32)
G = int(month.max()) + 1
ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731


def run(name, plan, n_in, n_out, n_read=None):
    """n_read: cells actually read when the operation touches only part of its input (dice)."""
    vals = eng.empty(n_in, DT)
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n_in, 0, CODE, 99, 1.0, eng.stream()))
    out = eng.empty(n_out, DT)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    gbs = ((n_in if n_read is None else n_read) + n_out) * ELEM / (ms * 1e-3) / 1e9
    print("%-46s %9.1f us %9.1f GB/s  %.3f  %s" % (name, ms * 1e3, gbs, gbs / 8000, plan.kernel_name), flush=True)


P = pkg.Plan
run("drillUp day -> month", P.drillup(DT, 0.0, "sum", shape, [G, 101, 271], [month, ident(101), ident(271)]), n, G * 101 * 271)
run("drillUp location -> 10 interleaved groups", P.drillup(DT, 0.0, "sum", shape, [3653, 10, 271], [ident(3653), (np.arange(101) % 10).astype(np.uint32), ident(271)]), n, 3653 * 10 * 271)
run("drillUp product -> all", P.drillup(DT, 0.0, "sum", shape, [3653, 101, 1], [ident(3653), ident(101), np.zeros(271, np.uint32)]), n, 3653 * 101)
sel = [np.arange(3653, dtype=np.int32), np.arange(0, 101, 3, dtype=np.int32), np.arange(271, dtype=np.int32)]
run("dice 34 of 101 locations (bytes = 2 x selected)", P.dice(DT, 0.0, shape, [3653, 34, 271], sel), n, 3653 * 34 * 271, 3653 * 34 * 271)
sel0 = [np.arange(0, 3653, 3, dtype=np.int32), np.arange(101, dtype=np.int32), np.arange(271, dtype=np.int32)]
run("dice every third day (bytes = 2 x selected)", P.dice(DT, 0.0, shape, [len(sel0[0]), 101, 271], sel0), n, len(sel0[0]) * 101 * 271, len(sel0[0]) * 101 * 271)
run("drillDown month -> day", P.drilldown(DT, 0.0, "sum", [G, 101, 271], shape, [month, ident(101), ident(271)]), G * 101 * 271, n)
run("reorder (product, location, day)", P.reorder(DT, 0.0, shape, [2, 1, 0]), n, n)
