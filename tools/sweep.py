#!/usr/bin/env python3
"""Developer tool: times drillUp over shapes/axes/methods on one GPU (HIP events on torch's stream)
and prints algorithmic GB/s, to find the shapes where a kernel is far from the HBM roofline.

  python tools/sweep.py [--quick]
"""
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
DT = os.environ.get("DTYPE", "float32")  # float32 | float64: the cell type of every case
ELEM, CODE = {"float32": (4, 2), "float64": (8, 3)}[DT]


def bench(plan_fn, n_in, n_out, elem=ELEM, with_mask=False, iters=30, reads=None):
    """reads: cells the operation has to READ (default: every input cell; `first` / `last` over a dense cube need one row
    per group only and the row kernel stops there)."""
    vals = eng.empty(n_in, DT)
    st = eng.empty(n_in, "int32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), st.data_ptr(), n_in, 0, CODE, 1234, 0.9 if with_mask else 1.0, eng.stream()))
    out = eng.empty(n_out, DT)
    ost = eng.empty(n_out, "int32")
    plan = plan_fn()
    args = (vals.data_ptr(), st.data_ptr() if with_mask else None, out.data_ptr(), ost.data_ptr() if with_mask else None, eng.stream())
    if os.environ.get("OLAP_SWEEP_PTRS"):  # where the buffers of this case lie (placement experiments)
        print("    in %#x  mask %#x  out %#x  out mask %#x" % (vals.data_ptr(), st.data_ptr(), out.data_ptr(), ost.data_ptr()), flush=True)
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nbytes = ((n_in if reads is None else reads) + n_out) * (elem + (4 if with_mask else 0))
    return ms, nbytes / (ms * 1e-3) / 1e9, plan.kernel_name


def ident(n):
    return np.arange(n, dtype=np.uint32)


rows = []
quick = "--quick" in sys.argv
shape = [10] * 8
n = 10 ** 8
for axis in range(8):
    new = list(shape)
    new[axis] = 1
    maps = [np.zeros(10, np.uint32) if i == axis else ident(10) for i in range(8)]
    for method in (["sum"] if quick else ["sum", "highest", "first"]):
        # (the row regime — axes 0-4 here — reads only the first set member of every group for `first`: one row of ten)
        first_rows = method == "first" and axis <= 4
        ms, gbs, k = bench(lambda: pkg.Plan.drillup(DT, 0.0, method, shape, new, maps), n, n // 10, reads=n // 10 if first_rows else None)
        rows.append(("[10]^8 axis%d->all %s%s" % (axis, method, " (reads 1 row of 10)" if first_rows else ""), ms, gbs, k))
ms, gbs, k = bench(lambda: pkg.Plan.drillup(DT, 0.0, "sum", shape, [1] + shape[1:], [np.zeros(10, np.uint32)] + [ident(10)] * 7), n, n // 10, with_mask=True)
rows.append(("[10]^8 axis0->all sum +mask", ms, gbs, k))
# config 5 shapes
s5 = [3652, 100, 274]
n5 = int(np.prod(s5))
day_to_month = (np.arange(3652) // 30.4375).astype(np.uint32)
G = int(day_to_month.max()) + 1
for method in ["sum", "average", "first", "last"]:
    picks = method in ("first", "last")  # one day of every month suffices on a dense cube
    ms, gbs, k = bench(lambda: pkg.Plan.drillup(DT, 0.0, method, s5, [G, 100, 274], [day_to_month, ident(100), ident(274)]), n5, G * 27400,
                       reads=G * 27400 if picks else None)
    rows.append(("C5 day->month %s%s" % (method, " (reads 1 day per month)" if picks else ""), ms, gbs, k))
city_to_country = (np.arange(100) // 10).astype(np.uint32)
ms, gbs, k = bench(lambda: pkg.Plan.drillup(DT, 0.0, "sum", s5, [3652, 10, 274], [ident(3652), city_to_country, ident(274)]), n5, 3652 * 10 * 274)
rows.append(("C5 city->country sum (3652 outer)", ms, gbs, k))
interleaved = (np.arange(100) % 10).astype(np.uint32)
ms, gbs, k = bench(lambda: pkg.Plan.drillup(DT, 0.0, "sum", s5, [3652, 10, 274], [ident(3652), interleaved, ident(274)]), n5, 3652 * 10 * 274)
rows.append(("C5 interleaved groups sum", ms, gbs, k))
# dice / reorder / drilldown
sel = [ident(10)] * 8
sel4 = list(sel)
sel4[4] = np.array([1, 4, 7], np.int32)
ms, gbs, k = bench(lambda: pkg.Plan.dice(DT, 0.0, shape, [10, 10, 10, 10, 3, 10, 10, 10], sel4), n, 3 * 10 ** 7)
rows.append(("dice dim4 3-of-10 (bytes=in+out)", ms, (2 * 3e7 * ELEM) / (ms * 1e-3) / 1e9, k))
sel1 = list(sel)
sel1[1] = np.array([3], np.int32)
ms, gbs, k = bench(lambda: pkg.Plan.dice(DT, 0.0, shape, [10, 1, 10, 10, 10, 10, 10, 10], sel1), n, 10 ** 7)
rows.append(("dice dim1 1-of-10", ms, (2 * 1e7 * ELEM) / (ms * 1e-3) / 1e9, k))
ms, gbs, k = bench(lambda: pkg.Plan.reorder(DT, 0.0, shape, [7, 6, 5, 4, 3, 2, 1, 0]), n, n)
rows.append(("reorder reverse [10]^8", ms, gbs, k))
ms, gbs, k = bench(lambda: pkg.Plan.reorder(DT, 0.0, shape, [1, 0, 2, 3, 4, 5, 6, 7]), n, n)
rows.append(("reorder swap dim0/1", ms, gbs, k))
ms, gbs, k = bench(lambda: pkg.Plan.reorder(DT, 0.0, [10000, 10000], [1, 0]), n, n)
rows.append(("reorder transpose [1e4,1e4]", ms, gbs, k))
ms, gbs, k = bench(lambda: pkg.Plan.reorder(DT, 0.0, [3652, 100, 274], [2, 1, 0]), n5, n5)
rows.append(("reorder C5 [3652,100,274] reversed", ms, gbs, k))
# load (in-memory.js:139-176): the other store's cells scattered into this one through per-dimension item maps
sel_id = [np.arange(10, dtype=np.int32)] * 8
ms, gbs, k = bench(lambda: pkg.Plan.load(DT, 0.0, 0.0, shape, shape, sel_id), n, n)
rows.append(("load [10]^8 identity item maps", ms, gbs, k))
perm4 = list(sel_id)
perm4[4] = np.array([3, 1, 4, 0, 9, 2, 6, 5, 8, 7], np.int32)
ms, gbs, k = bench(lambda: pkg.Plan.load(DT, 0.0, 0.0, shape, shape, perm4), n, n)
rows.append(("load [10]^8 items of dim4 remapped", ms, gbs, k))
perm7 = list(sel_id)
perm7[7] = np.array([3, 1, 4, 0, 9, 2, 6, 5, 8, 7], np.int32)
ms, gbs, k = bench(lambda: pkg.Plan.load(DT, 0.0, 0.0, shape, shape, perm7), n, n)
rows.append(("load [10]^8 items of dim7 (innermost) remapped", ms, gbs, k))
drop0 = list(sel_id)
drop0[0] = np.array([0, 1, 2, -1, 3, 4, -1, 5, 6, 7], np.int32)  # two of his items are unknown here; mine has 8 (+ untouched cells)
ms, gbs, k = bench(lambda: pkg.Plan.load(DT, 0.0, 0.0, [8] + shape[1:], shape, drop0), n, 8 * 10 ** 7)
rows.append(("load [10]^8 -> [8,10^7] two items dropped", ms, (1e8 + 8e7) * ELEM / (ms * 1e-3) / 1e9, k))
if "--only-reorder" in sys.argv:
    rows = [r for r in rows if r[0].startswith("reorder")]
if "--only-load" in sys.argv:
    rows = [r for r in rows if r[0].startswith("load")]
month_of_day = day_to_month
ms, gbs, k = bench(lambda: pkg.Plan.drilldown(DT, 0.0, "sum", [G, 100, 274], s5, [month_of_day, ident(100), ident(274)]), G * 27400, n5)
rows.append(("drillDown month->day", ms, gbs, k))
for r in rows:
    print("%-40s %9.1f us %9.1f GB/s  %.3f  %s" % (r[0], r[1] * 1e3, r[2], r[2] / 8000.0, r[3]))
