"""CPU-side checks of the C ABI: the library loads, exports every symbol include/olap_hip.h
declares, and validates arguments before it touches a device (same errors with and without a GPU)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_package

pkg = load_package()
capi = pkg.capi


def header_functions():
    text = open(os.path.join(ROOT, "include", "olap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(olap_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_and_binding_agree():
    declared = header_functions()
    assert declared, "no declarations found in include/olap_hip.h"
    assert sorted(capi.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for name in header_functions():
        assert hasattr(L, name), name
    assert L.olap_abi_version() == 2


def test_names_and_sizes():
    L = capi.lib()
    for name, code in capi.METHODS.items():
        assert L.olap_method_from_name(name.encode()) == code
    assert L.olap_method_from_name(None) == capi.METHODS["sum"]
    assert L.olap_method_from_name(b"median") == capi.ERR_UNSUPPORTED_METHOD
    assert capi.last_error() == "Unsupported aggregation method: median"
    for name, code in capi.DTYPES.items():
        assert L.olap_dtype_from_name(name.encode()) == code
        assert L.olap_dtype_size(code) == capi.DTYPE_SIZE[code]
    assert L.olap_dtype_from_name(b"float16") == capi.ERR_INVALID_TYPE
    assert capi.last_error() == "Invalid type"


def expect(code, message, fn, *args, **kw):
    with pytest.raises(pkg.OlapError) as ei:
        fn(*args, **kw)
    assert ei.value.code == code, (ei.value.code, str(ei.value))
    if message is not None:
        assert message in str(ei.value)


def test_argument_errors_precede_device_use():
    """These must be identical on the GPU box: validation happens before any HIP call."""
    expect(capi.ERR_INVALID_DEFAULT, "Invalid default value, only NaN and 0 are supported", pkg.HipStore, 4, "float32", 1.0)
    expect(capi.ERR_INVALID_TYPE, "Invalid type", pkg.HipStore, 4, "float16", 0.0)
    expect(capi.ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: median", pkg.Plan.drillup,
           "float32", 0.0, "median", [3], [1], [[0, 0, 0]])
    expect(capi.ERR_INDEX_RANGE, "outside the new dimension", pkg.Plan.drillup,
           "float32", 0.0, "sum", [3], [1], [[0, 0, 1]])
    expect(capi.ERR_INDEX_RANGE, "outside the old dimension", pkg.Plan.dice,
           "float32", 0.0, [3], [2], [[0, 3]])
    expect(capi.ERR_INVALID_ARGUMENT, "not a permutation", pkg.Plan.reorder, "float32", 0.0, [3, 2], [0, 0])
    expect(capi.ERR_INDEX_RANGE, "outside the old dimension", pkg.Plan.drilldown,
           "float32", 0.0, "sum", [1], [3], [[0, 0, 1]])
    expect(capi.ERR_INDEX_RANGE, "outside this store's dimension", pkg.Plan.load,
           "float32", 0.0, 0.0, [2], [2], [[0, 2]])


def test_batch_entry_points_validate_before_the_device():
    """olap_plan_run_batch / olap_store_drillup_batch (several measures of a cube in one call) refuse NULL lists and
    NULL members before anything touches a device."""
    L = capi.lib()
    assert L.olap_plan_run_batch(None, 2, None, None, None, None, None) == capi.ERR_INVALID_ARGUMENT
    assert "plan is NULL" in capi.last_error()
    assert L.olap_store_drillup_batch(2, None, None, 0, None, None, None, 0) == capi.ERR_INVALID_ARGUMENT
    hs, outs = (C.c_void_p * 2)(None, None), (C.c_void_p * 2)()
    assert L.olap_store_drillup_batch(2, hs, outs, 0, None, None, None, 0) == capi.ERR_INVALID_ARGUMENT
    assert "store 0 of the batch is NULL" in capi.last_error()
    assert L.olap_store_drillup_batch(0, hs, outs, 0, None, None, None, 0) == 0  # an empty cube: nothing to do
    methods = (C.c_int * 2)(0, 99)
    assert L.olap_store_drillup_multi(2, hs, methods, outs, 0, None, None, None) == capi.ERR_INVALID_ARGUMENT  # NULL stores come first
    assert L.olap_store_drillup_multi(2, None, methods, outs, 0, None, None, None) == capi.ERR_INVALID_ARGUMENT
    assert L.olap_store_drillup_multi(0, hs, methods, outs, 0, None, None, None) == 0
    assert L.olap_plan_run_batch_rules(None, 2, methods, None, None, None, None, None) == capi.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("K,G,inner,kind,dtype", [(1000, 10, 1, "mod", "float32"), (1000, 100, 1, "mod", "float32"), (100, 10, 10, "mod", "float32"),
                                                  (613, 0, 5, "ragged", "float32"), (37, 0, 3, "random", "int32"), (2048, 7, 1, "mod", "float64"),
                                                  (4096, 585, 1, "random", "uint32"), (96, 6, 32, "mod", "float32")])
def test_tile_placement_of_interleaved_groups(K, G, inner, kind, dtype):
    """Host side of the row-tile regime with interleaved groups (tile_perm_build): every cell of a row gets its own LDS
    cell, a group's members form ONE run in ascending member order, and — when the padding was affordable — the runs of
    32 consecutive reducing lanes start on 32 different LDS banks."""
    import numpy as np

    rng = np.random.default_rng(K + inner)
    if kind == "mod":
        amap = (np.arange(K) % G).astype(np.uint32)
    else:
        raw = np.minimum(rng.geometric(0.15, size=K) - 1, 20) if kind == "ragged" else rng.integers(0, max(2, K // 7), size=K)
        first = {}
        amap = np.array([first.setdefault(int(g), len(first)) for g in raw], dtype=np.uint32)
        G = int(amap.max()) + 1
    cell = np.zeros(K * inner, np.uint32)
    grp = np.zeros(2 * G, np.uint32)
    pitch = C.c_uint32()
    L = capi.lib()
    capi.check(L.olap_diag_tile_placement(capi.DTYPES[dtype], K, G, inner, amap.ctypes.data_as(capi._pu32), cell.ctypes.data_as(capi._pu32),
                                          grp.ctypes.data_as(capi._pu32), C.byref(pitch)))
    P = pitch.value
    assert P >= K and len(set(cell.tolist())) == K * inner and int(cell.max()) < P * inner
    size = np.bincount(amap, minlength=G)
    starts, ends = grp[0::2].astype(np.int64), grp[1::2].astype(np.int64)
    assert np.array_equal(ends - starts, size) and np.all(starts[1:] >= ends[:-1]) and ends[-1] <= P
    for g in range(G):  # the run of group g holds its members in ascending order, `inner` cells each
        members = np.nonzero(amap == g)[0]
        for r, k in enumerate(members):
            assert np.array_equal(cell[k * inner:(k + 1) * inner], (starts[g] + r) * inner + np.arange(inner))
    if P > K or np.all(starts == np.concatenate(([0], np.cumsum(size)[:-1]))) and inner >= 32:
        # lanes (row, group, i): bank of the first cell each reads
        rows = max(1, min(4, (16384 // capi.DTYPE_SIZE[capi.DTYPES[dtype]]) // (P * inner)))
        banks = [((r * P + starts[g]) * inner + i) % 32 for r in range(rows) for g in range(G) for i in range(inner)]
        for lo in range(0, max(1, len(banks) - 31), 32):
            window = banks[lo:lo + 32]
            assert len(set(window)) == len(window), (lo, window)
    cap = 16384 // capi.DTYPE_SIZE[capi.DTYPES[dtype]]
    assert L.olap_diag_tile_placement(capi.DTYPES[dtype], cap + 1, 1, 1, np.zeros(cap + 1, np.uint32).ctypes.data_as(capi._pu32), cell.ctypes.data_as(capi._pu32),
                                      grp.ctypes.data_as(capi._pu32), C.byref(pitch)) == capi.ERR_INVALID_ARGUMENT


def test_no_cpu_fallback_without_device():
    if capi.lib().olap_device_count() > 0:
        pytest.skip("a GPU is present")
    expect(capi.ERR_NO_DEVICE, "no CPU fallback", pkg.HipStore, 4, "float32", 0.0)
    expect(capi.ERR_NO_DEVICE, "no CPU fallback", pkg.Plan.drillup, "float32", 0.0, "sum", [3], [1], [[0, 0, 0]])


def test_product_does_not_reference_the_oracle():
    """The product tree must not import, link or open anything under oracle/."""
    bad = []
    for base, _dirs, files in os.walk(os.path.join(ROOT, "olap-in-memory_amd")):
        if os.sep + "build" in base or os.sep + "lib" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".py", ".js", ".cc", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle", text, flags=re.I):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def _run_dry_worker(extra_env):
    import subprocess
    import sys

    env = dict(os.environ, OLAP_PLAN_DRY="1", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_plan_dry_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=1500, env=env)
    assert r.returncode == 0 and "dry planning ok" in r.stdout, r.stdout[-6000:]


def test_planning_code_without_a_device():
    """OLAP_PLAN_DRY=1: every planner (drillUp regimes, dice, fused dice->drillUp, load, reorder forms, drillDown) builds
    its plan with the tables in host memory; running such a plan is refused (no CPU fallback)."""
    _run_dry_worker({})


def test_planning_code_under_asan_and_ubsan():
    """the host-side planning code (CSR, tile cuts, remap / brick / transpose tables) under
    AddressSanitizer + UndefinedBehaviorSanitizer — the same worker against the sanitizer build of libolapgpu
    (host code instrumented, device code as usual; GPU sanitizers are unavailable on this pool)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("olap_build_for_asan", os.path.join(ROOT, "olap-in-memory_amd", "build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    lib = build.build_lib_asan()
    _run_dry_worker({"OLAP_LIBOLAPGPU": lib, "LD_PRELOAD": build.asan_runtime(), "ASAN_OPTIONS": "detect_leaks=0",
                     "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})


def test_sharded_entry_points_validate_before_the_device():
    """The multi-GPU entry points: host-only arithmetic works without a device, argument errors come first, and
    without a device nothing falls back (OLAP_ERR_NO_DEVICE)."""
    import ctypes as C

    from olap_in_memory_amd import sharded

    L = capi.lib()
    assert sharded.partition_rows(10, 3) == [0, 4, 7, 10]
    expect(capi.ERR_INVALID_ARGUMENT, "world must be >= 1", sharded.partition_rows, 10, 0)
    expect(capi.ERR_INVALID_ARGUMENT, "sharded:", sharded.dice_bounds, [0, 5, 10], [3, 3])
    expect(capi.ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method", sharded.recipe, "float32", 0.0, 7)
    expect(capi.ERR_INVALID_TYPE, "Invalid type", lambda: capi.check(L.olap_shard_recipe_get(9, 0, 0, C.byref(capi.ShardRecipe()))))
    h = C.c_void_p()
    assert L.olap_sharded_store_create(C.byref(h), None, 1, (C.c_uint32 * 1)(4), 2, 0, None) == capi.ERR_INVALID_ARGUMENT
    assert L.olap_shard_drillup_create(C.byref(h), None, 2, 0, 0, 1, None, None, None, None, 0, 1) == capi.ERR_INVALID_ARGUMENT
    assert L.olap_comm_init_all(C.byref(h), (C.c_int * 2)(0, 0), 0) == capi.ERR_INVALID_ARGUMENT
    assert L.olap_comm_init_detached(C.byref(h), 2, 5, 0) == capi.ERR_INVALID_ARGUMENT
    if L.olap_device_count() == 0:
        expect(capi.ERR_NO_DEVICE, "no CPU fallback", sharded.Comm.init_all, [0, 0])
        expect(capi.ERR_NO_DEVICE, "no CPU fallback", sharded.Comm.detached, 2, 0, 0)


def test_rccl_that_cannot_be_loaded_is_an_error_not_a_crash():
    """the loader read dlerror() twice — the second call returns NULL — and built a std::string from
    it: a host without librccl crashed inside olap_comm_unique_id instead of reporting OLAP_ERR_NO_DEVICE, which is what
    Comm.from_process_group and bench.py's labelled gloo rehearsal rely on.  OLAP_RCCL_LIB names the library to bind."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_package\n"
        "pkg = load_package(); capi = pkg.capi\n"
        "from olap_in_memory_amd.sharded import Comm\n"
        "for attempt in range(2):\n"  # the failure is remembered, not re-raised as a crash the second time
        "    try:\n"
        "        Comm.unique_id()\n"
        "        print('no error'); break\n"
        "    except capi.OlapError as e:\n"
        "        print('code', e.code, str(e))\n"
    ) % (os.path.join(ROOT, "tests"), ROOT)
    env = dict(os.environ, OLAP_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("code")]
    assert len(lines) == 2 and all(("code %d " % capi.ERR_NO_DEVICE) in l and "cannot load RCCL" in l and "/nonexistent/librccl.so.1" in l for l in lines), r.stdout
