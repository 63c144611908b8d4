"""Worker of tests/test_capi_nogpu.py::test_planning_code_*: builds a few hundred launch plans with
OLAP_PLAN_DRY=1 (no device: the index tables land in host memory, nothing is launched) so that libolapgpu's
host-side planning — CSR of the roll-up maps, row / tile / group-tile / reduce regime cuts, remap tables, brick
and two-axis transpose descriptions, drillDown tables — runs on the CPU, normally under ASan + UBSan
(olap-in-memory_amd/build.py:build_lib_asan).  Shapes include every regime boundary the GPU tests use."""
import os
import sys

import numpy as np

assert os.environ.get("OLAP_PLAN_DRY") == "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402

pkg = load_package()
P = pkg.Plan
rng = np.random.default_rng(2024)
ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
TYPES = [("float32", 0.0), ("float32", float("nan")), ("int32", 0.0), ("uint32", float("nan")), ("float64", 0.0)]
METHODS = ["sum", "average", "highest", "lowest", "first", "last", "product"]
seen = {}


def note(plan, n_in, n_out):
    assert plan.in_cells == n_in and plan.out_cells == n_out, (plan.kernel_name, plan.in_cells, n_in, plan.out_cells, n_out)
    seen[plan.kernel_name] = seen.get(plan.kernel_name, 0) + 1
    plan.destroy()


# ---- drillUp: every regime (rows / tile / gtile / flat / reduce / split / generic)
SHAPES = [([10] * 8, 0), ([10] * 8, 7), ([10] * 8, 4), ([3652, 100, 274], 0), ([27400, 3652], 1), ([900, 3652, 30], 1), ([1000, 1000, 100], 1),
          ([10 ** 8], 0), ([10 ** 5, 1000], 1), ([10 ** 6, 100], 0), ([3653, 101, 271], 1), ([3653, 101, 271], 2), ([7, 9, 513], 1), ([1, 100000], 1),
          ([300, 4000], 1), ([5, 70000], 1), ([40000, 200], 0), ([3001, 3333, 10], 1), ([2, 3, 4096], 2), ([12, 3, 171], 0), ([0, 5], 0), ([5, 0], 1)]
SANITIZED = "LD_PRELOAD" in os.environ  # under ASan the 10^8-member roll-up (400 MB of tables) only costs time
for lens, axis in SHAPES:
    K = lens[axis]
    if SANITIZED and K > 10 ** 6:
        continue
    for kind in ("all", "runs", "interleaved", "random"):
        if K == 0:
            gmap, G = np.zeros(0, np.uint32), 1
        elif kind == "all":
            gmap, G = np.zeros(K, np.uint32), 1
        elif kind == "runs":
            G = max(1, min(K, 1 + K // 31))
            gmap = np.minimum(np.arange(K) // 31, G - 1).astype(np.uint32)
        elif kind == "interleaved":
            G = max(1, min(K, 10))
            gmap = (np.arange(K) % G).astype(np.uint32)
        else:
            G = max(1, min(K, 7))
            gmap = rng.integers(0, G, size=K).astype(np.uint32)
        new = list(lens)
        new[axis] = G
        maps = [gmap if i == axis else ident(l) for i, l in enumerate(lens)]
        for t, d in TYPES[:3] if K > 10 ** 6 else TYPES:
            m = METHODS[int(rng.integers(0, 7))]
            note(P.drillup(t, d, m, lens, new, maps), int(np.prod(lens)), int(np.prod(new)))
for _ in range(40):  # maps on several dimensions: the generic form
    nd = int(rng.integers(2, 5))
    lens = [int(x) for x in rng.integers(1, 9, size=nd)]
    new = [int(rng.integers(1, l + 1)) for l in lens]
    maps = [rng.integers(0, g, size=l).astype(np.uint32) for l, g in zip(lens, new)]
    note(P.drillup("float32", 0.0, METHODS[int(rng.integers(0, 7))], lens, new, maps), int(np.prod(lens)), int(np.prod(new)))

# ---- dice / fused dice -> drillUp / load
for lens in ([10] * 8, [3653, 101, 271], [6, 5, 8], [7, 16], [2, 3, 4096], [3, 40, 129], [300, 8]):
    for _ in range(4):
        sel = []
        for l in lens:
            mode = int(rng.integers(0, 4))
            if mode == 0:
                sel.append(np.arange(l, dtype=np.int32))
            elif mode == 1:
                sel.append(np.arange(0, l, 3, dtype=np.int32))
            elif mode == 2:
                sel.append(rng.permutation(l)[: max(1, l // 2)].astype(np.int32))
            else:
                s = rng.integers(-1, l, size=int(rng.integers(1, l + 2))).astype(np.int32)
                sel.append(s)
        new = [len(s) for s in sel]
        for t, d in TYPES:
            note(P.dice(t, d, lens, new, sel), int(np.prod(lens)), int(np.prod(new)))
        axis = int(rng.integers(0, len(lens)))
        out = list(new)
        out[axis] = 1
        maps = [np.zeros(l, np.uint32) if i == axis else ident(l) for i, l in enumerate(new)]
        note(P.dice_drillup("float32", 0.0, METHODS[int(rng.integers(0, 7))], lens, new, out, sel, maps), int(np.prod(lens)), int(np.prod(out)))
        h2m = [np.where(rng.random(len(s)) < 0.8, np.minimum(np.arange(len(s)), l - 1), -1).astype(np.int32) for s, l in zip(sel, lens)]
        note(P.load("float32", 0.0, float("nan"), lens, new, h2m), int(np.prod(new)), int(np.prod(lens)))

# ---- reorder: gather, bricks (16-byte and scalar), the two-axis transpose
for lens in ([10] * 8, [10] * 6, [12, 7, 20], [64, 48], [1000, 1000], [50, 100, 1000], [37, 53], [6, 1, 5, 4], [3, 250, 9, 30], [20, 30, 40], [3653, 101, 271],
             [130, 3, 131], [5, 300, 7, 260], [2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2]):
    for _ in range(5):
        perm = [int(x) for x in rng.permutation(len(lens))]
        for t, d in TYPES:
            note(P.reorder(t, d, lens, perm), int(np.prod(lens)), int(np.prod(lens)))

# ---- drillDown: row form, line-aligned form, two-pass, integer spreading, distributions
for old, fan, axis in (([120, 100, 274], 30, 0), ([12, 101, 271], 31, 0), ([7, 5, 20], 3, 1), ([4, 6], 5, 1), ([3, 2, 9, 4], 2, 2), ([1], 3, 0)):
    new = list(old)
    new[axis] = old[axis] * fan
    child = np.repeat(np.arange(old[axis]), fan).astype(np.uint32)
    if old[axis] > 1 and rng.random() < 0.5:
        child = np.sort(rng.integers(0, old[axis], size=new[axis])).astype(np.uint32)
    maps = [child if i == axis else ident(l) for i, l in enumerate(old)]
    for t, d in TYPES:
        for m in ("sum", "average"):
            note(P.drilldown(t, d, m, old, new, maps), int(np.prod(old)), int(np.prod(new)))
    note(P.drilldown("float32", 0.0, "sum", old, new, maps, rng.random(int(np.prod(new)))), int(np.prod(old)), int(np.prod(new)))

# a dry plan never launches
p = P.drillup("float32", 0.0, "sum", [4, 4], [1, 4], [np.zeros(4, np.uint32), ident(4)])
try:
    p.run(16, None, 16, None)
    raise AssertionError("a dry plan must refuse to run")
except pkg.OlapError as e:
    assert e.code == pkg.capi.ERR_NO_DEVICE and "no CPU fallback" in str(e)
print("plans built:", sum(seen.values()), "kernels:", sorted(seen))
for k in ("drillup_rows_kernel", "drillup_tile_kernel", "drillup_gtile_kernel", "drillup_flat_kernel", "drillup_generic_kernel", "transpose_xy_kernel",
          "reorder_brick4_kernel", "reorder_brick_kernel", "gather(reorder)", "drilldown_rows_kernel"):
    assert any(k in name for name in seen), k
print("dry planning ok")
