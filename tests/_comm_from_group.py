"""Worker for tests/test_sharded_gloo.py::test_comm_from_torch_process_group: what bench.py does at N > 1 to get its RCCL
communicator — a torch.distributed "nccl" group carries rank 0's unique id (Comm.from_process_group) — on a one-rank
group, followed by one sharded step on that communicator."""
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

pkg = load_package()
from olap_in_memory_amd import capi  # noqa: E402
from olap_in_memory_amd.sharded import Comm, ShardedStore  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29549")
torch.cuda.set_device(0)
capi.check(capi.lib().olap_set_device(0))
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
comm = Comm.from_process_group(dist, 0)
assert comm.transport == "rccl" and comm.world == 1
lens = [12, 50, 40]
s = ShardedStore(comm, lens, "float32", 0.0).fill_seeded(3, 0.7)
op = s.plan_drillup_dim0(np.zeros(12, np.uint32), 1, "sum", placement=capi.PLACE_SCATTER, depth=2)
vals, stat = s.step_inputs()
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    op.step(vals, stat, [stream])
op.wait([stream])
torch.cuda.synchronize()
got, _, first = op.result_host(0)
v, keep = config_cube(24000, 3, 0.7)
want = v.astype(np.float64).reshape(12, 2000).sum(0).astype(np.float32)
assert first == 0 and np.allclose(got, want, rtol=1e-6, atol=0)
dist.barrier()
dist.destroy_process_group()
print("comm from group ok")
