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
    phases = ""
    if hasattr(L, "olap_diag_xy_probe") and "xy" in plan.kernel_name:
        # a library built with -DOLAP_XY_PROBE (OLAP_LIBOLAPGPU=...): mean time from a workgroup's start to its tables
        # being ready, its tile being in LDS and its last store being issued, and the workgroups resident per CU
        import ctypes as C

        plan.run(*args)
        torch.cuda.synchronize()
        nb = 1 << 17
        buf = (C.c_ulonglong * (nb * 8))()
        L.olap_diag_xy_probe.restype, L.olap_diag_xy_probe.argtypes = C.c_int, [C.c_void_p, C.c_ulonglong]
        assert L.olap_diag_xy_probe(buf, nb * 8) == 0
        q = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
        q = q[q[:, 3] > q[:, 0]]
        span = (q[:, 3].max() - q[:, 0].min()) * 0.01
        phases = "  tables %.2f us, tile in LDS %.2f us, stores issued %.2f us, resident/CU %.1f (first %d workgroups)" % (
            (q[:, 1] - q[:, 0]).mean() * 0.01, (q[:, 2] - q[:, 0]).mean() * 0.01, (q[:, 3] - q[:, 0]).mean() * 0.01,
            (q[:, 3] - q[:, 0]).sum() * 0.01 / span / 256, len(q))
    print("%-20s perm %-10s %8.1f us  %.3f  %s%s" % (shape, perm, us, 2 * n * 4 / (us * 1e-6) / 8e12, plan.kernel_name, phases), flush=True)
    del vals, out
