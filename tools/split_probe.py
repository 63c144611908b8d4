#!/usr/bin/env python3
"""Developer tool: roll-ups with few output cells, long groups and rows wider than 128 cells (the split regime)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
from olap_in_memory_amd.sharded import HipEngine
eng = HipEngine("cuda:0"); L = pkg.lib()
def run(name, lens, axis, iters=100):
    n = int(np.prod(lens)); new = list(lens); new[axis] = 1
    maps = [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    vals = eng.empty(n, "float32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 1234, 1.0, eng.stream()))
    out = eng.empty(n // lens[axis], "float32")
    plan = pkg.Plan.drillup("float32", 0.0, "sum", lens, new, maps)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(10): plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): plan.run(*args)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    print("%-26s %7.2f us  %.3f  %s" % (name, us, (n + n // lens[axis]) * 4 / us / 1e6 / 8.0, plan.kernel_name), flush=True)
run("[1e5,1000]->[1,1000]", [10**5, 1000], 0)
run("[1e4,1e4]->[1,1e4]", [10**4, 10**4], 0)
run("[1000,1e5]->[1,1e5]", [1000, 10**5], 0)
run("[1e6,100]->[1,100]", [10**6, 100], 0)
run("[4e5,250]->[1,250]", [4*10**5, 250], 0)
run("[2e5,500]->[1,500]", [2*10**5, 500], 0)
run("[10,1e4,1000]->[10,1,1000]", [10, 10**4, 1000], 1)
