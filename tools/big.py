#!/usr/bin/env python3
"""Developer tool: BASELINE configs[3] (10^9 cells) on ONE GPU, both shapes of SURVEY §8(e): the
literal [10]*9 and the shard-friendly [320,5,5,5,5,5,5,10,20]; drillUp(sum) of dimension 0 -> all.
This is the single-GPU time the 8-GPU sharded run is compared with."""
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
shapes = ([10] * 9, [320, 5, 5, 5, 5, 5, 5, 10, 20])
if "--per-rank" in sys.argv:  # the slab one rank of the 8-GPU run reduces
    shapes = ([40, 5, 5, 5, 5, 5, 5, 10, 20],)
for shape in shapes:
    n = int(np.prod(shape))
    n_out = n // shape[0]
    vals = eng.empty(n, "float32")
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, 2, 20240807, 1.0, eng.stream()))
    out = eng.empty(n_out, "float32")
    new = [1] + shape[1:]
    maps = [np.zeros(shape[0], np.uint32)] + [np.arange(l, dtype=np.uint32) for l in shape[1:]]
    plan = pkg.Plan.drillup("float32", 0.0, "sum", shape, new, maps)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    gbs = (n + n_out) * 4 / (ms * 1e-3) / 1e9
    # spot check against float64 numpy on a slice of the output
    k = 1000
    head = vals[: 0].cpu()  # (keeps the import of torch honest)
    cols = torch.stack([vals[r * n_out: r * n_out + k] for r in range(shape[0])]).double().sum(0).float().cpu().numpy()
    assert np.array_equal(cols, out[:k].cpu().numpy()), "mismatch against float64 column sums"
    print("%-34s %9.1f us  %8.1f GB/s  %.3f  %.3g cells/s  %s" % (shape if len(set(shape)) > 1 else "[10]*9", ms * 1e3, gbs, gbs / 8000, n / (ms * 1e-3), plan.kernel_name), flush=True)
    del vals, out
    torch.cuda.empty_cache()
