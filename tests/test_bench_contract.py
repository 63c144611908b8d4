"""bench.py's one-line JSON contract (what the driver parses), checked on a real run with few steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key, kind in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(d[key], kind), (key, d[key])
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.3 < roof["frac"] < 1.0
    assert roof["traffic"] is None or (roof["traffic"] > 0 and "NOT measured in this run" in roof["traffic_source"])
    cpu = d["cpu_baseline"]
    assert "1e8 cells" in cpu["sample"] and "sample ratio 1" in cpu["sample"]  # the baseline runs the headline configuration itself
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    # whole-job throughput and the per-step time describe the same run
    cells = d["config"]["cells_per_gpu"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_sharded_code_path_on_one_rank():
    """`--force-sharded` drives every line of bench.py's N > 1 path — sharded store, RCCL communicator made with
    olap_comm_init_rank, pipelined reduce-scatter steps, the serial-steps figure, the literal [10]^9 shape — with a
    one-rank communicator, so API misuse shows up on a 1-GPU box and not on the driver's 8-GPU run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--steps", "4", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]  # RCCL's version banner must not reach standard output
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["transport"] == "rccl" and d["config"]["collective"] == "reduce_scatter"
    assert d["config"]["shape"] == [320, 5, 5, 5, 5, 5, 5, 10, 20] and d["config"]["cells_per_gpu"] == 10 ** 9
    assert d["serial_steps"]["ms_per_step"] > 0 and d["literal_shape"]["rows_per_rank"] == [10]
    assert abs(d["value"] - 1e9 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 0.3 < d["roofline"]["frac"] < 1.0
    # the line stands on its own in a scaling sweep: the N = 1 point of the same 10^9 cells, what travels, every rank's kernel
    base = d["scaling_base"]
    assert base["n_gpus"] == 1 and base["cells"] == 10 ** 9 and base["cells_per_s"] > 1e11 and base["shape"] == d["config"]["shape"]
    assert d["config"]["partial_type"] == "float64" and d["config"]["partial_bytes_per_rank"] == 8 * 10 ** 9 // 320
    assert len(d["roofline"]["kernel_ms_per_rank"]) == 1 and isinstance(d["cpu_baseline_ref"], str)
    assert d["literal_shape"]["partial_bytes_per_rank"] == 8 * 10 ** 8


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 --rehearse` from a bare interpreter: the script launches its two ranks itself (child
    processes, before this process touches the GPU) and relays rank 0's line (the ranks share the one GPU over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["transport"] == "detached" and d["config"]["cells_per_gpu"] == 5 * 10 ** 7


@pytest.mark.gpu
def test_bench_refuses_fewer_gpus_than_asked_for():
    """`--gpus N` on a node with fewer than N GPUs fails loudly (non-zero, a message on standard error, nothing on standard output)
    instead of running a smaller job under the larger label."""
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing to run fewer ranks" in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-500:])
