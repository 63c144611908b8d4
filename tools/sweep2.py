#!/usr/bin/env python3
"""Developer tool: drillUp on awkward shapes (long rows with inner=1, tiny outputs, ...)."""
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


def run(name, lens, axis, amap, method="sum", iters=20):
    n = int(np.prod(lens))
    G = int(np.max(amap)) + 1
    new = list(lens)
    new[axis] = G
    maps = [np.asarray(amap, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    vals = eng.empty(n, "float32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 1234, 1.0, eng.stream()))
    n_out = n // lens[axis] * G
    out = eng.empty(n_out, "float32")
    plan = pkg.Plan.drillup("float32", 0.0, method, lens, new, maps)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    gbs = (n + n_out) * 4 / (ms * 1e-3) / 1e9
    print("%-44s %9.1f us %9.1f GB/s  %.3f  %s" % (name, ms * 1e3, gbs, gbs / 8000, plan.kernel_name), flush=True)


month = (np.arange(3652) // 30.4375).astype(np.uint32)
run("[27400,3652] day innermost -> month", [27400, 3652], 1, month)
run("[27400,3652] day innermost -> all", [27400, 3652], 1, np.zeros(3652))
run("[274,100,3652] day innermost -> month", [274, 100, 3652], 2, month)
run("[1e6,100] -> [1,100]", [10 ** 6, 100], 0, np.zeros(10 ** 6))
run("[100,1e6] -> [100,1]", [100, 10 ** 6], 1, np.zeros(10 ** 6))
run("[1e4,1e4] axis1 -> all", [10 ** 4, 10 ** 4], 1, np.zeros(10 ** 4))
run("[1e5,1000] axis1 -> all", [10 ** 5, 1000], 1, np.zeros(1000))
run("[1e5,1000] axis1 -> 10 interleaved", [10 ** 5, 1000], 1, np.arange(1000) % 10)
run("[1e8] -> [1]", [10 ** 8], 0, np.zeros(10 ** 8))
run("[1e7,10] axis0 -> all", [10 ** 7, 10], 0, np.zeros(10 ** 7))
# few outputs, rows wider than 128 cells: the split regime
run("[1e4,1e4] axis0 -> all", [10 ** 4, 10 ** 4], 0, np.zeros(10 ** 4))
run("[1e5,1000] axis0 -> all", [10 ** 5, 1000], 0, np.zeros(10 ** 5))
run("[4e5,250] axis0 -> all (rows of 250 cells)", [4 * 10 ** 5, 250], 0, np.zeros(4 * 10 ** 5))
run("[1000,1000,100] axis1 -> 100 groups", [1000, 1000, 100], 1, np.arange(1000) % 100)
run("[1000,1000,100] axis1 highest", [1000, 1000, 100], 1, np.arange(1000) % 100, "highest")
run("[1000,1000,100] axis1 product", [1000, 1000, 100], 1, np.arange(1000) % 100, "product")
run("[100,1e6] sum NaN-default n/a", [100, 10 ** 6], 0, np.zeros(100))
run("[1000,1000,100] axis1 -> 100 contiguous groups", [1000, 1000, 100], 1, np.arange(1000) // 10)
run("[900,3652,30] day in the middle -> month", [900, 3652, 30], 1, month)
run("[274,3652,100] day in the middle -> month", [274, 3652, 100], 1, month)
run("[274,3652,100] day in the middle -> month, first", [274, 3652, 100], 1, month, "first")
# odd trailing extents: no 16 B alignment between rows
run("[10,3001,3333] axis0 -> all (odd inner)", [10, 3001, 3333], 0, np.zeros(10))
run("[3001,10,3333] axis1 -> all (odd inner)", [3001, 10, 3333], 1, np.zeros(10))
run("[3001,3333,10] axis1 -> 11 groups (inner 10)", [3001, 3333, 10], 1, np.arange(3333) // 303)
# rows whose slots fill the last workgroup of a row only partly
run("[400,100,1200] axis1 -> 10 interleaved (300 slots/row)", [400, 100, 1200], 1, np.arange(100) % 10)
run("[700,100,700] axis1 -> 10 interleaved (175 slots/row)", [700, 100, 700], 1, np.arange(100) % 10)
run("[3653,101,271] axis1 -> 10 interleaved (271 4-byte slots)", [3653, 101, 271], 1, np.arange(101) % 10)
run("[1800,100,540] axis1 -> 10 interleaved (135 slots/row)", [1800, 100, 540], 1, np.arange(100) % 10)
