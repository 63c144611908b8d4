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
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    # whole-job throughput and the per-step time describe the same run
    cells = d["config"]["cells_per_gpu"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
