#!/usr/bin/env python3
"""Developer tool: host->device copy of a 400 MB pageable buffer, plain hipMemcpy against
hipHostRegister + hipMemcpy + hipHostUnregister (is page-locking a one-off buffer worth its cost?)."""
import ctypes as C
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
n = 100_000_000
host = np.ones(n, np.float32)
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), C.c_size_t(n * 4)) == 0
p = host.ctypes.data_as(C.c_void_p)


def t(label, fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    print("%-58s %7.1f ms  %6.1f GB/s" % (label, best * 1e3, n * 4 / best / 1e9), flush=True)


def plain():
    assert hip.hipMemcpy(dev, p, C.c_size_t(n * 4), 1) == 0


def registered():
    assert hip.hipHostRegister(p, C.c_size_t(n * 4), 0) == 0
    assert hip.hipMemcpy(dev, p, C.c_size_t(n * 4), 1) == 0
    assert hip.hipHostUnregister(p) == 0


t("hipMemcpy from pageable memory", plain)
t("hipHostRegister + hipMemcpy + hipHostUnregister", registered)
assert hip.hipHostRegister(p, C.c_size_t(n * 4), 0) == 0
t("hipMemcpy from already registered memory", plain)
hip.hipHostUnregister(p)
back = np.empty(n, np.float32)
q = back.ctypes.data_as(C.c_void_p)
t("device -> pageable host", lambda: hip.hipMemcpy(q, dev, C.c_size_t(n * 4), 2))
