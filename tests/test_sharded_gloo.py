"""world_size-2 rehearsal of the dim0-sharded drillUp (olap-in-memory_amd/sharded.py) over gloo.

CPU run: the oracle stands in for the local kernels (injected engine), so what is under test is
the row partition, the row sub-maps, the collective choice and the rank-ordered combine.
GPU run (-m gpu): the same worker with the HIP engine, two ranks sharing the one GPU of the box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import config_cube
from oracle.oracle import OracleStore

HERE = os.path.dirname(os.path.abspath(__file__))


def run_workers(engine, tmp_path, world=2):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + world), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_worker.py"), engine, out], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=300)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return [json.load(open("%s.%d" % (out, r))) for r in range(world)]


def check(results):
    lens = [7, 6, 10]
    row_map = np.array([0, 1, 0, 2, 1, 0, 2], np.uint32)
    maps = [row_map, np.arange(6, dtype=np.uint32), np.arange(10, dtype=np.uint32)]
    for frac in (1.0, 0.4):
        v, _ = config_cube(420, 77, frac)
        o = OracleStore(420, "float32", 0.0)
        o.set_data(v.astype(np.float64))
        for method in ("sum", "average", "highest", "lowest", "first", "last", "product"):
            ev, _ = o.drill_up(lens, [3, 6, 10], maps, method).typed()
            got = np.full(180, np.nan)
            for res in results:
                r = res["%s_%s" % (method, frac)]
                got[r["range"][0]:r["range"][1]] = r["values"]
            if method in ("sum", "average", "product"):  # float32 partials combined across ranks: 1e-5 relative (north star)
                assert np.allclose(got, ev, rtol=1e-5, atol=0), method
            else:
                assert np.array_equal(got.astype(np.float32), ev), method
        ev, _ = o.drill_up(lens, [3, 6, 10], maps, "sum").typed()
        got = np.full(180, np.nan)
        for res in results:
            r = res["pipelined_%s" % frac]
            got[r["range"][0]:r["range"][1]] = r["values"]
        assert np.allclose(got, ev, rtol=1e-5, atol=0), "pipelined sum"
        e2, _ = o.drill_up(lens, [7, 6, 2], [np.arange(7, dtype=np.uint32), np.arange(6, dtype=np.uint32),
                                             (np.arange(10) % 2).astype(np.uint32)], "sum").typed()
        got = np.full(84, np.nan)
        for res in results:
            r = res["axis2_%s" % frac]
            got[r["range"][0]:r["range"][1]] = r["values"]
        assert np.array_equal(got.astype(np.float32), e2)

        def gathered(key, n):
            out = np.full(n, np.nan)
            for res in results:
                r = res["%s_%s" % (key, frac)]
                out[r["range"][0]:r["range"][1]] = r["values"]
            return out.astype(np.float32)

        ident = lambda l: np.arange(l, dtype=np.int32)  # noqa: E731
        e3, _ = o.dice(lens, [7, 4, 3], [ident(7), np.array([4, 0, -1, 2], np.int32), np.array([9, 8, 1], np.int32)]).typed()
        assert np.array_equal(gathered("dice12", 84), e3)
        e4, _ = o.drill_down(lens, [7, 6, 30], [np.arange(7, dtype=np.uint32), np.arange(6, dtype=np.uint32),
                                                  np.repeat(np.arange(10), 3).astype(np.uint32)], "sum").typed()
        assert np.array_equal(gathered("down2", 1260), e4)
        picked = o.dice(lens, [4, 6, 10], [np.array([1, 2, 4, 6], np.int32), ident(6), ident(10)])
        e5, _ = picked.typed()
        assert np.array_equal(gathered("rows", 240), e5)
        e6, _ = picked.drill_up([4, 6, 10], [4, 1, 10], [np.arange(4, dtype=np.uint32), np.zeros(6, np.uint32), np.arange(10, dtype=np.uint32)], "sum").typed()
        assert np.array_equal(gathered("rows_then_sum", 40), e6)


def test_partition_rows():
    from conftest import load_package
    load_package()
    from olap_in_memory_amd.sharded import partition_rows
    assert partition_rows(10, 8) == [0, 2, 4, 5, 6, 7, 8, 9, 10]
    assert partition_rows(320, 8) == list(range(0, 321, 40))
    assert partition_rows(3, 4) == [0, 1, 2, 3, 3]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_drillup_gloo_cpu(tmp_path, world):
    """7 rows over 2 ranks (4 + 3) and over 3 ranks (3 + 2 + 2): ragged partitions."""
    check(run_workers("oracle", tmp_path, world))


@pytest.mark.gpu
def test_sharded_drillup_gloo_gpu(tmp_path):
    check(run_workers("hip", tmp_path))


@pytest.mark.gpu
def test_rccl_code_path_single_rank():
    """The exact torch.distributed / RCCL calls bench.py makes at N > 1 (reduce_scatter_tensor sync and
    async, all_reduce, all_gather_into_tensor), on a one-rank nccl group: catches API misuse that the
    gloo rehearsal cannot (device tensors, stream semantics) before the driver's 8-GPU run."""
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_single_rank.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "rccl single-rank ok" in r.stdout, r.stdout[-4000:]
