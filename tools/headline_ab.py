#!/usr/bin/env python3
"""Developer tool: 300-launch means of the headline shapes (one process per library: OLAP_LIBOLAPGPU selects another
build, e.g. olap-in-memory_amd/lib_prev/libolapgpu.so) — run alternately a few times inside ONE gpurun call, the only
way two builds compare to better than the box-to-box spread."""
import datetime
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
tag = "prev" if os.environ.get("OLAP_LIBOLAPGPU") else "this"


def run(name, lens, axis, amap, method="sum", iters=300):
    n = int(np.prod(lens))
    G = int(np.max(amap)) + 1
    new = list(lens)
    new[axis] = G
    maps = [np.asarray(amap, np.uint32) if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    vals = eng.empty(n, "float32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 1234, 1.0, eng.stream()))
    out = eng.empty(n // lens[axis] * G, "float32")
    plan = pkg.Plan.drillup("float32", 0.0, method, lens, new, maps)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(30):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    print("%s %-34s %7.2f us  %s" % (tag, name, a.elapsed_time(b) / iters * 1e3, plan.kernel_name), flush=True)
    del vals, out


d0 = datetime.date(2010, 1, 1)
month = np.array([(d0 + datetime.timedelta(days=i)).month - 1 + 12 * ((d0 + datetime.timedelta(days=i)).year - 2010) for i in range(3652)])
run("[10]^8 dim0->all sum", [10] * 8, 0, np.zeros(10))
run("[10]^8 dim0->all highest", [10] * 8, 0, np.zeros(10), "highest")
run("[10]^8 dim5->all sum", [10] * 8, 5, np.zeros(10))
run("C5 day->month sum", [3652, 100, 274], 0, month)
run("C5 day->month last", [3652, 100, 274], 0, month, "last")
run("[1e5,1000]->10 interleaved", [100000, 1000], 1, np.arange(1000) % 10)
run("[1000,1000,100] interleaved", [1000, 1000, 100], 1, np.arange(1000) % 100)
