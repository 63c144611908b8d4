#!/usr/bin/env python3
"""Developer tool: the few-outputs reduce regime by number of units (OLAP_REDUCE_UNITS=n: groups x segments <= n; unset: the
plan's own choice)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
from olap_in_memory_amd.sharded import HipEngine
eng = HipEngine("cuda:0"); L = pkg.lib()
def run(name, lens, axis, iters=200):
    n = int(np.prod(lens)); new = list(lens); new[axis] = 1
    maps = [np.zeros(l, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    vals = eng.empty(n, "float32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 1234, 1.0, eng.stream()))
    out = eng.empty(n // lens[axis], "float32")
    plan = pkg.Plan.drillup("float32", 0.0, "sum", lens, new, maps)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(20): plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): plan.run(*args)
    b.record(); torch.cuda.synchronize()
    print("units %5s %-22s %7.2f us" % (os.environ.get("OLAP_REDUCE_UNITS", "auto"), name, a.elapsed_time(b) / iters * 1e3), flush=True)
run("[1e6,100]->[1,100]", [10**6, 100], 0)
run("[1e7,10]->[1,10]", [10**7, 10], 0)
run("[1e8]->[1]", [10**8], 0)
run("[100,1e6]->[100,1]", [100, 10**6], 1)
run("[1e5,1000]->[1e5,1]", [10**5, 1000], 1)
run("[300,3e5]->[300,1]", [300, 300000], 1)
run("[3,3e7]->[3,1]", [3, 3 * 10**7], 1)
run("[1e6,40,2]->[1,40,2]", [10**6, 40, 2], 0)
run("[2000,5e4]->[2000,1]", [2000, 50000], 1)
