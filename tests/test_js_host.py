"""Runs the Node.js host's own test files (tests/js/*.js) under node.

host_test.js needs no GPU (dimensions, calendar, formatters, argument errors);
gpu_test.js drives the full Cube API through the N-API addon on the device;
reference_cases.js restates every case of the reference's own test-suite (test/*.js) one for one —
its dimension-only cases run on the CPU tier (`--host`), all 115 on the GPU."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NODE = shutil.which("node")
ADDON = os.path.join(os.path.dirname(HERE), "olap-in-memory_amd", "lib", "olapgpu.node")


def run_node(script, *args, env=None):
    r = subprocess.run([NODE, os.path.join(HERE, "js", script), *args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-6000:]
    return r.stdout


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_js_host_logic_without_gpu():
    out = run_node("host_test.js")
    assert "0 failed" in out


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_reference_suite_dimension_cases_without_gpu():
    out = run_node("reference_cases.js", "--host")
    assert "38 passed, 0 failed" in out


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_addon_is_built():
    assert os.path.exists(ADDON), "olapgpu.node missing: run __graft_entry__.build()"


@pytest.mark.gpu
def test_js_cube_api_on_gpu():
    assert NODE is not None, "node is expected on the GPU box (same image)"
    out = run_node("gpu_test.js")
    assert "0 failed" in out


@pytest.mark.gpu
def test_reference_suite_on_gpu():
    """All 115 cases of the reference's test/*.js (benchmark file aside), same inputs and literals."""
    out = run_node("reference_cases.js")
    assert "115 passed, 0 failed" in out, out[-6000:]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_reference_suite_on_gpu_sharded(devices):
    """The same 115 cases with every stored measure SPLIT along its outermost dimension (OLAP_DEVICES names one
    device two / three times: the shards exchange by direct reads, RCCL refuses two ranks on one device).
    Cube.drillUp of the sharded dimension runs as partial + one collective behind the same store call
    (src/cube.js:1012-1020); what the shards cannot answer in place is gathered first."""
    out = run_node("reference_cases.js", env={"OLAP_DEVICES": devices})
    assert "115 passed, 0 failed" in out, out[-6000:]


@pytest.mark.gpu
def test_js_sharded_cube_on_gpu():
    out = run_node("sharded_test.js")
    assert "0 failed" in out, out[-6000:]
