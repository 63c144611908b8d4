#!/usr/bin/env python3
"""Developer tool: olap_store_totals (getNestedObject(measure, withTotals), src/cube.js:421-440) on cubes whose extended
cube does not fit one workgroup's LDS: launches, bytes read from HBM as a multiple of the cube, and the time of the
call (which includes the blocking copy of the float64 export to the host).  OLAP_TOTALS_NO_GROUPS=1 gives the
round-2 form (scatter + one launch per dimension + export) for comparison."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
CASES = [("[10]^6", [10] * 6), ("4^10", [4] * 10), ("[100,100,100]", [100, 100, 100]), ("C5 [3652,100,274]", [3652, 100, 274]), ("[10]^7", [10] * 7)]
for name, lens in CASES:
    n = int(np.prod(lens))
    ext = int(np.prod([l + 1 for l in lens]))
    g = pkg.HipStore(n, "float32", 0.0)
    g.fill_seeded(7, 1.0) if hasattr(g, "fill_seeded") else g.fill(1.0)
    g.totals(lens, ["sum"] * len(lens))
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        e, st, launches, nbytes = g.totals(lens, ["sum"] * len(lens))
    dt = (time.perf_counter() - t0) / reps
    print("%-22s cube %10d cells, extended %10d (x %.2f): %d launches, bytes read = %.2f x the cube, %.2f ms per call (of which the %.0f MB export copy); total of all cells %.6g"
          % (name, n, ext, ext / n, launches, nbytes / (n * 4), dt * 1e3, ext * 12 / 1e6, e[-1]), flush=True)
    del g
