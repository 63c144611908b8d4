#!/usr/bin/env python3
"""Developer tool: two builds of libolapgpu loaded into ONE process and timed on the SAME buffers, alternating —
separates "this binary is slower" from "this process got slower memory"."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = os.path.join(ROOT, "olap-in-memory_amd", "lib", "libolapgpu.so")
B = os.path.join(ROOT, "olap-in-memory_amd", "lib_prev", "libolapgpu.so")
libs = {"new": C.CDLL(A), "old": C.CDLL(B)}
for L in libs.values():
    L.olap_reorder_plan.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.olap_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.olap_plan_kernel_name.restype = C.c_char_p
    L.olap_plan_kernel_name.argtypes = [C.c_void_p]
N = 10 ** 8
a = torch.rand(N, dtype=torch.float32, device="cuda:0")
b = torch.empty(N, dtype=torch.float32, device="cuda:0")
CASES = [([10] * 8, list(range(7, -1, -1))), ([10000, 10000], [1, 0]), ([3652, 100, 274], [2, 1, 0])]
for shape, perm in CASES:
    plans = {}
    for k, L in libs.items():
        h = C.c_void_p()
        ol = np.asarray(shape, np.uint32)
        pp = np.asarray(perm, np.int32)
        rc = L.olap_reorder_plan(C.byref(h), 0, 0, len(shape), ol.ctypes.data_as(C.POINTER(C.c_uint32)), pp.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0, rc
        plans[k] = h
    for rnd in range(3):
        for k, L in libs.items():
            for _ in range(3):
                L.olap_plan_run(plans[k], a.data_ptr(), None, b.data_ptr(), None, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.olap_plan_run(plans[k], a.data_ptr(), None, b.data_ptr(), None, None)
            e1.record()
            torch.cuda.synchronize()
            print("%-24s %s round %d  %8.1f us  %s" % (shape, k, rnd, e0.elapsed_time(e1) / 10 * 1e3, L.olap_plan_kernel_name(plans[k]).decode()), flush=True)
