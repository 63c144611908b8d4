#!/usr/bin/env python3
"""Developer tool: 2-D transposes of 10^8 cells with different row strides on the two sides, to see which side of the
two-axis transpose is sensitive to the distance between consecutive rows (pages)."""
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
CASES = [([100000, 1000], [1, 0]), ([31623, 3162], [1, 0]), ([10000, 10000], [1, 0]), ([3162, 31623], [1, 0]), ([1000, 100000], [1, 0]), ([100, 1000000], [1, 0]),
         ([3652, 27400], [1, 0]), ([3652, 100, 274], [1, 2, 0]), ([3652, 100, 274], [2, 1, 0]), ([3652, 100, 274], [2, 0, 1]), ([1000, 1000, 100], [2, 1, 0]),
         ([1000, 1000, 100], [1, 2, 0]), ([100, 1000, 1000], [2, 1, 0])]
for shape, perm in CASES:
    n = int(np.prod(shape))
    vals = eng.empty(n, "float32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 99, 1.0, eng.stream()))
    out = eng.empty(n, "float32")
    plan = pkg.Plan.reorder("float32", 0.0, shape, perm)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    print("%-20s perm %-10s %8.1f us  %.3f  %s" % (shape, perm, us, 2 * n * 4 / (us * 1e-6) / 8e12, plan.kernel_name), flush=True)
    del vals, out
