"""Builds the native parts of the package in-tree (the built files travel to the GPU box):

  lib/libolapgpu.so   HIP kernels + C ABI (include/olap_hip.h), hipcc --offload-arch=gfx950
  lib/olapgpu.node    N-API addon for the Node.js host (only when node headers are present)

hipcc cross-compiles gfx950 without a GPU, so this runs in the authoring container too.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
NAPI = os.path.join(HERE, "napi")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

KERNEL_UNITS = ["olap_kernels_f32.hip", "olap_kernels_f64.hip", "olap_kernels_i32.hip", "olap_kernels_u32.hip",
                "olap_capi.hip", "olap_sharded.hip", "olap_transpose.hip", "olap_totals.hip", "olap_order.hip"]
HEADERS = ["olap_device.hpp", "olap_kernels.hpp", "olap_internal.hpp", os.path.join(ROOT, "include", "olap_hip.h")]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def lib_path():
    return os.path.join(LIB, "libolapgpu.so")


def addon_path():
    return os.path.join(LIB, "olapgpu.node")


def build_lib(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-ffp-contract=off", "-I", os.path.join(ROOT, "include")]
    jobs = []
    objs = []
    for unit in KERNEL_UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJ, unit.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    target = lib_path()
    if force or jobs or _newer(target, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", target] + objs + ["-ldl"])
    return target


def asan_lib_path():
    return os.path.join(HERE, "lib_asan", "libolapgpu.so")


def build_lib_asan(force=False):
    """The same sources with AddressSanitizer + UBSan on the HOST side only (device code is compiled as usual:
    GPU sanitizers are not available on this pool) -> lib_asan/libolapgpu.so.  Used with OLAP_PLAN_DRY=1 to run the
    planning code under the sanitizers on a machine without a GPU (tests/test_capi_nogpu.py)."""
    out_dir, obj_dir = os.path.join(HERE, "lib_asan"), os.path.join(HERE, "build_asan")
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    flags = ["--offload-arch=" + ARCH, "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=off", "-fsanitize=address,undefined",
             "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include")]
    jobs, objs = [], []
    for unit in KERNEL_UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(obj_dir, unit.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(_run, jobs))
    target = asan_lib_path()
    if force or jobs or _newer(target, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-o", target] + objs + ["-ldl"])
    return target


def asan_runtime():
    """Path of clang's shared ASan runtime (to LD_PRELOAD into python)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return _run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"]).strip()


def node_include_dir():
    for d in ("/usr/include/node", "/usr/local/include/node"):
        if os.path.exists(os.path.join(d, "node_api.h")):
            return d
    return None


def build_addon(force=False):
    """N-API addon; returns None when this machine has no node headers."""
    inc = node_include_dir()
    src = os.path.join(NAPI, "olap_napi.cc")
    if inc is None or not os.path.exists(src):
        return None
    os.makedirs(LIB, exist_ok=True)
    target = addon_path()
    if force or _newer(target, [src, os.path.join(ROOT, "include", "olap_hip.h")]):
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-DNAPI_VERSION=6", "-DNODE_GYP_MODULE_NAME=olapgpu", "-I", inc, "-I",
              os.path.join(ROOT, "include"), src, "-o", target, "-L", LIB, "-lolapgpu", "-Wl,-rpath,$ORIGIN"])
    return target


def build_all(force=False, verbose=False):
    lib = build_lib(force=force, verbose=verbose)
    addon = build_addon(force=force)
    return lib, addon


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
