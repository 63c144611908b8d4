"""Worker for tests/test_sharded_gloo.py::test_config4_sharded_on_one_gpu: BASELINE configs[3]'s 10^9-cell cube sharded
over the ranks of a gloo group that share the one GPU of the box (detached communicator: gloo carries the payloads,
RCCL refuses two ranks on a device).  One sharded drillUp(sum) of dimension 0 -> all through the product's
olap_sharded_store + olap_shard_drillup; every rank checks slices of ITS block of the result against float64 column
sums recomputed from the position-addressed generator."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from golden_util import mulberry32_at  # noqa: E402

pkg = load_package()
from olap_in_memory_amd import capi, sharded  # noqa: E402

shape = [int(x) for x in sys.argv[1].split(",")]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
capi.check(capi.lib().olap_set_device(0))
comm = sharded.Comm.detached(world, rank, 0)
SEED = 20240807
store = sharded.ShardedStore(comm, shape, "float32", 0.0).fill_seeded(SEED, 1.0)
n = int(np.prod(shape))
n_out = n // shape[0]
bounds = store.bounds
assert bounds == sharded.partition_rows(shape[0], world)
op = store.plan_drillup_dim0(np.zeros(shape[0], np.uint32), 1, "sum", placement=capi.PLACE_SCATTER)
assert op.local_cells(0) == (bounds[rank + 1] - bounds[rank]) * n_out and op.out_cells == n_out
vals, stat = store.step_inputs()
op.local(0, vals[0], None)
sharded.exchange_over_process_group(op, dist)
op.finish(0)
got, st, first = op.result_host(0)
per = -(-n_out // world)
assert first == min(rank * per, n_out) and got.size == max(0, min(per, n_out - first)), (first, got.size)
for lo in (0, got.size // 2, max(0, got.size - 500)):
    k = min(500, got.size - lo)
    if k <= 0:
        continue
    cols = np.zeros(k)
    for r in range(shape[0]):
        cells = np.arange(r * n_out + first + lo, r * n_out + first + lo + k, dtype=np.uint64)
        cols += (0.5 + mulberry32_at(SEED, 2 * cells + 1)).astype(np.float32).astype(np.float64)
    # per-rank partial sums are rounded to Float32 before the exchange: 1e-5 relative (north star); here <= 2 ulp
    assert np.allclose(got[lo:lo + k], cols, rtol=1e-6, atol=0), (rank, lo, got[lo:lo + 4], cols[:4])
dist.barrier()
if rank == 0:
    print("sharded 10^9 ok", shape, "world", world, "rows per rank", [b - a for a, b in zip(bounds, bounds[1:])])
dist.destroy_process_group()
