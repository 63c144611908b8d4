"""Worker for tests/test_sharded_gloo.py::test_rccl_code_path_single_rank: a one-rank RCCL group on
cuda:0 running every collective branch of olap-in-memory_amd/sharded.py (sum via reduce-scatter,
pipelined sum, average via (sum,count), highest via all-gather + combine)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from golden_util import config_cube  # noqa: E402
from oracle.oracle import OracleStore  # noqa: E402

pkg = load_package()
from olap_in_memory_amd.sharded import HipEngine, ShardedStore  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = HipEngine("cuda:0")
lens = [12, 50, 40]
row_map = (np.arange(12) % 3).astype(np.uint32)
maps = [row_map, np.arange(50, dtype=np.uint32), np.arange(40, dtype=np.uint32)]
s = ShardedStore(lens, "float32", 0.0, 0, 1, eng).fill_seeded(3, 0.7)
vals, _ = config_cube(24000, 3, 0.7)
o = OracleStore(24000, "float32", 0.0)
o.set_data(vals.astype(np.float64))
for method in ("sum", "average", "highest", "first"):
    op = s.plan_drillup_dim0(row_map, 3, method, always_collective=True)
    got = op.step().cpu().numpy()
    ev, _ = o.drill_up(lens, [3, 50, 40], maps, method).typed()
    assert np.allclose(got, ev, rtol=1e-6, atol=0), method
op = s.plan_drillup_dim0(row_map, 3, "sum", always_collective=True)
outs = [op.step_pipelined() for _ in range(5)]
op.flush()
torch.cuda.synchronize()
ev, _ = o.drill_up(lens, [3, 50, 40], maps, "sum").typed()
assert np.allclose(outs[-1].cpu().numpy(), ev, rtol=1e-6, atol=0) and np.allclose(outs[-2].cpu().numpy(), ev, rtol=1e-6, atol=0)
dist.barrier()
dist.destroy_process_group()
print("rccl single-rank ok")
