#!/usr/bin/env python3
"""Developer tool: mid-size `inner` shapes (flat vs LDS tile regime)."""
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
    print("%-44s %9.1f us %9.1f GB/s  %.3f" % (name, ms * 1e3, gbs, gbs / 8000), flush=True)


run("[10]^8 axis5 (inner 100, K 10)", [10] * 8, 5, np.zeros(10))
run("[1e5,10,100] axis1 (inner 100)", [10 ** 5, 10, 100], 1, np.zeros(10))
run("[1e5,25,40] axis1 5 groups (inner 40)", [10 ** 5, 25, 40], 1, np.arange(25) % 5)
run("[1e6,10,10] axis1 (inner 10)", [10 ** 6, 10, 10], 1, np.zeros(10))
run("[1e5,50,20] axis1 (inner 20)", [10 ** 5, 50, 20], 1, np.zeros(50))
run("[1e4,100,64] axis1 10 groups (inner 64)", [10 ** 4, 100, 64], 1, np.arange(100) // 10)
run("[2e4,12,400] axis1 (inner 400)", [2 * 10 ** 4, 12, 400], 1, np.zeros(12))
