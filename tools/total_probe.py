#!/usr/bin/env python3
"""Developer tool: the store total (collapse of an additive measure) on 10^8 and 10^6 cells; per call, including the
blocking read of the result (OLAP_TOTAL_BLOCKS=n: workgroups of the first stage)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (torch's HIP runtime first)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
tag = os.environ.get("OLAP_TOTAL_BLOCKS", "prev" if os.environ.get("OLAP_LIBOLAPGPU") else "auto")
for n in (10 ** 8, 10 ** 6):
    s = pkg.HipStore(n, "float32", 0.0)
    s.set_data(np.random.default_rng(1).integers(0, 5, size=n).astype(np.float32))
    want = float(s.get_data().astype(np.float64).sum())
    for _ in range(5):
        got = s.total
    assert got == want, (got, want)
    t0 = time.perf_counter()
    for _ in range(100):
        s.total
    print("blocks %5s  total of %.0e cells  %8.1f us per call" % (tag, n, (time.perf_counter() - t0) / 100 * 1e6), flush=True)
