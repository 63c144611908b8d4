#!/usr/bin/env python3
"""Developer tool: drillDown month->day (float32, sum) for several inner extents, to separate the
cost of rows that do not start on a 128-byte line from the rest.

  python tools/dd_bench.py
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
DT = os.environ.get("DTYPE", "float32")  # float32 | float64
ELEM, CODE = {"float32": (4, 2), "float64": (8, 3)}[DT]
days = np.arange(np.datetime64("2010-01-01"), np.datetime64("2020-01-01"))
months = days.astype("datetime64[M]").astype(np.int64)
month_of_day = (months - months[0]).astype(np.uint32)
G, K = int(month_of_day.max()) + 1, len(days)

for inner in (27400, 27392, 27648, 8192, 65536):
    n_in, n_out = G * inner, K * inner
    vals = eng.empty(n_in, DT)
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n_in, 0, CODE, 7, 1.0, eng.stream()))
    out = eng.empty(n_out, DT)
    plan = pkg.Plan.drilldown(DT, 0.0, "sum", [G, inner], [K, inner], [month_of_day, np.arange(inner, dtype=np.uint32)])
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 30
    gbs = (n_in + n_out) * ELEM / (ms * 1e-3) / 1e9
    print("inner=%6d  %8.1f us %8.1f GB/s  %.3f  %s" % (inner, ms * 1e3, gbs, gbs / 8000.0, plan.kernel_name), flush=True)
    del vals, out
