#!/usr/bin/env python3
"""Developer tool: every case of tools/regime_probe.py x every rule, one cell type per run (DTYPE=float32|float64), in ONE
process — the way to find a regime x rule x cell-type combination that fell off a cliff (the sweeps roll most shapes up
with `sum` over Float32 only).  Prints microseconds per launch and the fraction of 8 TB/s counted over the whole cube
(`first` / `last` may read less than that in the row regime: fractions above 1 are theirs)."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("regime_probe", os.path.join(ROOT, "tools", "regime_probe.py"))
rp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rp)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from olap_in_memory_amd.sharded import HipEngine  # noqa: E402

eng = HipEngine("cuda:0")
L = pkg.lib()
dtype = os.environ.get("DTYPE", "float32")
size, code = {"float32": (4, 2), "float64": (8, 3)}[dtype]
iters = int(os.environ.get("ITERS", "20"))
only = set(sys.argv[1:])
METHODS = ["sum", "average", "highest", "first", "last", "product"]
print("%-18s %s" % (dtype, "  ".join("%-16s" % m for m in METHODS)))
for name, (lens, axis, mk) in rp.CASES.items():
    if only and name not in only:
        continue
    amap = np.asarray(mk(lens[axis]), np.uint32)
    n = int(np.prod(lens))
    new = list(lens)
    new[axis] = int(amap.max()) + 1
    maps = [amap if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    vals = eng.empty(n, dtype)
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, code, 1234, 1.0, eng.stream()))
    n_out = n // lens[axis] * new[axis]
    out = eng.empty(n_out, dtype)
    cells = []
    kernel = ""
    for method in METHODS:
        plan = pkg.Plan.drillup(dtype, 0.0, method, lens, new, maps)
        args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
        for _ in range(2):
            plan.run(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            plan.run(*args)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        frac = (n + n_out) * size / (us * 1e-6) / 8e12
        cells.append("%7.1f us %5.3f%s" % (us, frac, "!" if frac < 0.6 else " "))
        kernel = plan.kernel_name
        del plan
    print("%-18s %s  %s" % (name, "  ".join(cells), kernel), flush=True)
    del vals, out
    torch.cuda.empty_cache()
