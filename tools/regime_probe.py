#!/usr/bin/env python3
"""Developer tool: ONE drillUp case per process (name on the command line), so that
`rocprofv3 --kernel-trace --stats -- python3 tools/regime_probe.py <case>` splits its time by kernel
(main kernel / fold of the partials) — the regimes that sit below 0.70 of the HBM peak.
Without a case name: lists them.  OLAP_* planning knobs apply (they are read when the plan is built)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    "sq_axis0": ([10 ** 4, 10 ** 4], 0, lambda K: np.zeros(K)),
    "tall1000_axis0": ([10 ** 5, 1000], 0, lambda K: np.zeros(K)),
    "tall250_axis0": ([4 * 10 ** 5, 250], 0, lambda K: np.zeros(K)),
    "tall100_axis0": ([10 ** 6, 100], 0, lambda K: np.zeros(K)),
    "tall10_axis0": ([10 ** 7, 10], 0, lambda K: np.zeros(K)),
    "all_cells": ([10 ** 8], 0, lambda K: np.zeros(K)),
    "wide_axis1": ([100, 10 ** 6], 1, lambda K: np.zeros(K)),
    "day_innermost": ([27400, 3652], 1, lambda K: (np.arange(K) // 30.4375).astype(np.uint32)),
    "axis5": ([10 ** 5, 10, 100], 1, lambda K: np.zeros(K)),
    "axis6": ([10 ** 6, 10, 10], 1, lambda K: np.zeros(K)),
    "axis7": ([10 ** 7, 10], 1, lambda K: np.zeros(K)),
    "mid_month30": ([900, 3652, 30], 1, lambda K: (np.arange(K) // 30.4375).astype(np.uint32)),
    "mid_month100": ([274, 3652, 100], 1, lambda K: (np.arange(K) // 30.4375).astype(np.uint32)),
    "groups11": ([3001, 3333, 10], 1, lambda K: np.arange(K) // 303),
    "tile_interleaved": ([10 ** 5, 1000], 1, lambda K: np.arange(K) % 10),
    "rows1000_all": ([10 ** 5, 1000], 1, lambda K: np.zeros(K)),
    "c5_product": ([3653, 101, 271], 2, lambda K: np.zeros(K)),
    "flat100": ([1000, 1000, 100], 1, lambda K: np.arange(K) % 100),
    "flat100_random": ([1000, 1000, 100], 1, lambda K: np.random.default_rng(5).permutation(K) % 100),
    "flat100_blocks": ([1000, 1000, 100], 1, lambda K: (np.arange(K) // 5) % 100),
    "c5_country": ([3652, 100, 274], 1, lambda K: np.arange(K) // 10),
    "c5_interleaved": ([3652, 100, 274], 1, lambda K: np.arange(K) % 10),
    "narrow540": ([1800, 100, 540], 1, lambda K: np.arange(K) % 10),
    "narrow540c": ([1800, 100, 540], 1, lambda K: np.arange(K) // 5),
    "narrow1200c": ([400, 100, 1200], 1, lambda K: np.arange(K) // 2),
    "narrow700": ([700, 100, 700], 1, lambda K: np.arange(K) % 10),
    "axis4": ([10 ** 4, 10, 1000], 1, lambda K: np.zeros(K)),
    "axis3": ([10 ** 3, 10, 10 ** 4], 1, lambda K: np.zeros(K)),
    "axis4_1024": ([10 ** 4, 10, 1024], 1, lambda K: np.zeros(K)),
    "axis1": ([10, 10, 10 ** 6], 1, lambda K: np.zeros(K)),
    "axis2": ([100, 10, 10 ** 5], 1, lambda K: np.zeros(K)),
    "c5_month": ([3652, 100, 274], 0, lambda K: (np.arange(K) // 30.4375).astype(np.uint32)),
    "odd_axis0": ([10, 3001, 3333], 0, lambda K: np.zeros(K)),
    "odd_axis1": ([3001, 10, 3333], 1, lambda K: np.zeros(K)),
    "odd_loc": ([3653, 101, 271], 1, lambda K: np.arange(K) % 10),
    "headline": ([10] * 8, 0, lambda K: np.zeros(K)),
}


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in CASES:
        print("cases:", " ".join(CASES))
        return
    from __graft_entry__ import load_package
    pkg = load_package()
    from olap_in_memory_amd.sharded import HipEngine
    eng = HipEngine("cuda:0")
    L = pkg.lib()
    lens, axis, mk = CASES[sys.argv[1]]
    method = sys.argv[2] if len(sys.argv) > 2 else "sum"
    iters = int(os.environ.get("ITERS", "30"))
    amap = np.asarray(mk(lens[axis]), np.uint32)
    n = int(np.prod(lens))
    new = list(lens)
    new[axis] = int(amap.max()) + 1
    maps = [amap if i == axis else np.arange(l, dtype=np.uint32) for i, l in enumerate(lens)]
    dtype = os.environ.get("DTYPE", "float32")  # float32 | float64
    size, code = {"float32": (4, 2), "float64": (8, 3)}[dtype]
    vals = eng.empty(n, dtype)
    pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, n, 0, code, 1234, 1.0, eng.stream()))
    n_out = n // lens[axis] * new[axis]
    out = eng.empty(n_out, dtype)
    plan = pkg.Plan.drillup(dtype, 0.0, method, lens, new, maps)
    args = (vals.data_ptr(), None, out.data_ptr(), None, eng.stream())
    for _ in range(3):
        plan.run(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        plan.run(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    gbs = (n + n_out) * size / (ms * 1e-3) / 1e9
    print("%-18s %-8s %9.1f us %9.1f GB/s  %.3f  %s" % (sys.argv[1], method, ms * 1e3, gbs, gbs / 8000, plan.kernel_name), flush=True)


if __name__ == "__main__":
    main()
