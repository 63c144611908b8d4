#!/usr/bin/env python3
"""Developer tool: the same transpose plan timed on several freshly allocated buffer pairs of one process — is the
fast / slow split of the transposes a property of the memory a buffer happens to get?"""
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
N = 10 ** 8
plan = pkg.Plan.reorder("float32", 0.0, [10000, 10000], [1, 0])
plan2 = pkg.Plan.reorder("float32", 0.0, [10] * 8, list(range(7, -1, -1)))


def timed(p, args, reps=10):
    for _ in range(3):
        p.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        p.run(*args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


keep = []
sizes = [N * 4, N * 4 + 4096, N * 4 + (2 << 20), 1 << 29, (1 << 29) + 4096, N * 4, N * 4]
for i, sz in enumerate(sizes):
    a = torch.empty(sz, dtype=torch.uint8, device="cuda:0")
    b = torch.empty(sz, dtype=torch.uint8, device="cuda:0")
    a[: N * 4].view(torch.float32).uniform_()
    keep.append((a, b))
    args = (a.data_ptr(), None, b.data_ptr(), None, eng.stream())
    print("pair %d size %10d  in %#x (%% 2MiB = %7d)  out %#x (%% 2MiB = %7d)   2-D %7.1f us   reversed %7.1f us" % (
        i, sz, a.data_ptr(), a.data_ptr() % (2 << 20), b.data_ptr(), b.data_ptr() % (2 << 20), timed(plan, args), timed(plan2, args)), flush=True)
# mixed pairs: input of pair i, output of pair j
for i, j in [(0, 1), (1, 0), (0, 3), (3, 0), (2, 5), (5, 2)]:
    args = (keep[i][0].data_ptr(), None, keep[j][1].data_ptr(), None, eng.stream())
    print("in of %d, out of %d   2-D %7.1f us   reversed %7.1f us" % (i, j, timed(plan, args), timed(plan2, args)), flush=True)
