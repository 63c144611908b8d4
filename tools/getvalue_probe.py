#!/usr/bin/env python3
"""Developer tool: latency of a single-cell read (olap_store_get_value) with and without a primary mask."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
tag = "prev" if os.environ.get("OLAP_LIBOLAPGPU") else "this"
for t, d in (("float32", 0.0), ("uint32", float("nan"))):
    s = pkg.HipStore(1000, t, d)
    s.set_data_f64(np.arange(1000, dtype=np.float64))
    for _ in range(50): s.get_value(7)
    t0 = time.perf_counter()
    for i in range(2000): s.get_value(i % 1000)
    print("%s getValue %-8s %6.1f us per call  (value %s)" % (tag, t, (time.perf_counter() - t0) / 2000 * 1e6, s.get_value(7)), flush=True)
