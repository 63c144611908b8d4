#!/usr/bin/env python3
"""Developer tool: same-process A/B of launch-time knobs on one drillUp case (tools/regime_probe.py's cases).
   python3 tools/ab_probe.py <case> [method] -- VAR=a,b [VAR2=c,d ...]
Every combination of the listed values is timed in alternating rounds (5 rounds x 100 launches each), so the box-to-box
and process-to-process spread (2-3 %) does not hide a real difference.  Only knobs read at LAUNCH time work here
(OLAP_ROWS_DEPTH, OLAP_XCD_ORDER, ...); the ones read once per process or at plan time need tools/regime_probe.py."""
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv0 = sys.argv[:]
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
head, knobs = args[:split], args[split + 1:]
sys.argv = [sys.argv[0]]  # regime_probe's main() must not run a case on import
import importlib.util
spec = importlib.util.spec_from_file_location("regime_probe", os.path.join(ROOT, "tools", "regime_probe.py"))
rp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rp)

from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
from olap_in_memory_amd.sharded import HipEngine  # noqa: E402

eng = HipEngine("cuda:0")
L = pkg.lib()
lens, axis, mk = rp.CASES[head[0]]
method = head[1] if len(head) > 1 else "sum"
amap = np.asarray(mk(lens[axis]), np.uint32)
n = int(np.prod(lens))
new = list(lens)
new[axis] = int(amap.max()) + 1
maps = [amap if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
vals = eng.empty(n, "float32")
pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 1234, 1.0, eng.stream()))
n_out = n // lens[axis] * new[axis]
out = eng.empty(n_out, "float32")
plan = pkg.Plan.drillup("float32", 0.0, method, lens, new, maps)
run_args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
names = [k.split("=")[0] for k in knobs]
values = [k.split("=")[1].split(",") for k in knobs]
combos = list(itertools.product(*values)) or [()]
times = {c: [] for c in combos}
for rnd in range(6):
    for c in combos:
        for k, v in zip(names, c):
            if v == "-":
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for _ in range(5):
            plan.run(*run_args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            plan.run(*run_args)
        b.record()
        torch.cuda.synchronize()
        if rnd:
            times[c].append(a.elapsed_time(b) * 10.0)  # us per launch
print("%s %s  %s" % (head[0], method, plan.kernel_name))
for c in combos:
    t = times[c]
    print("  %-40s mean %7.1f us  min %7.1f  max %7.1f   %.3f of 8 TB/s" % (
        " ".join("%s=%s" % (k.replace("OLAP_", ""), v) for k, v in zip(names, c)), sum(t) / len(t), min(t), max(t), (n + n_out) * 4 / (sum(t) / len(t) * 1e-6) / 8e12))
