#!/usr/bin/env python3
"""Developer tool: does the RELATIVE placement of the buffers a kernel streams at the same time change its speed?
One big allocation is carved by hand; the second stream (the mask of a masked drillUp, the destination of a
transpose) is moved by `delta` bytes against a 2 MiB-aligned base and the same plan is timed at every delta."""
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
MB2 = 2 << 20
N = 10 ** 8
slab = torch.empty(4 * (N * 4 + 64 * MB2), dtype=torch.uint8, device="cuda:0")
base = (slab.data_ptr() + MB2 - 1) // MB2 * MB2
STRIDE = (N * 4 + MB2 - 1) // MB2 * MB2 + 16 * MB2  # slots of 2 MiB-aligned starts, 16 MiB of slack each


def timed(plan, args, reps=10):
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


DELTAS = [0, 256, 1024, 4096, 4096 + 256, 16384, 65536, 65536 + 4096, 262144, 1 << 20, (1 << 20) + 4096, MB2 + 65536, 4 * MB2, 8 * MB2 + 4096]
if len(sys.argv) > 1:
    DELTAS = [int(x) for x in sys.argv[1].split(",")]
P = pkg.Plan
ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
pkg.capi.check(L.olap_fill_seeded(base, None, N, 0, 2, 99, 1.0, eng.stream()))
torch.cuda.synchronize()

cases = []
shape = [10] * 8
maps = [np.zeros(10, np.uint32)] + [ident(10)] * 7
cases.append(("drillUp [10]^8 axis0->all sum + mask (second stream: the mask)", P.drillup("float32", float("nan"), "sum", shape, [1] + [10] * 7, maps), "mask"))
cases.append(("reorder [10]^8 reversed (second stream: the destination)", P.reorder("float32", 0.0, shape, list(range(7, -1, -1))), "out"))
cases.append(("reorder [1e4,1e4] transposed", P.reorder("float32", 0.0, [10000, 10000], [1, 0]), "out"))
cases.append(("reorder C5 [3652,100,274] reversed", P.reorder("float32", 0.0, [3652, 100, 274], [2, 1, 0]), "out"))
off0 = base - slab.data_ptr()
for name, plan, what in cases:
    print(name, plan.kernel_name, flush=True)
    for d in DELTAS:
        if what == "mask":
            o = off0 + STRIDE + d
            slab[o:o + 4 * N].view(torch.int32).fill_(2)  # every cell set
            args = (base, base + STRIDE + d, base + 2 * STRIDE, base + 3 * STRIDE, eng.stream())
        else:
            args = (base, None, base + STRIDE + d, None, eng.stream())
        torch.cuda.synchronize()
        us = timed(plan, args)
        print("   delta %10d B  %8.1f us" % (d, us), flush=True)
