#!/usr/bin/env python3
"""Developer tool: the kernels that sit below 0.7 of the HBM roofline, each
launched REPS times between marker launches so that a rocprofv3 pass can be cut into cases.

  python tools/pmc_probe.py                 # timings only (HIP events), prints one line per case
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 tools/pmc_probe.py --quiet
  ... one pass per counter set (FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)
  python tools/pmc_probe.py --summarize r02 # gpurun_out/pmc_* + pmc_cases.json -> profiles/traffic_<tag>_kernels.json (+ a copy in gpurun_out/)

Markers: one olap_diag_read_ceiling launch (kernel `diag_read_kernel`) before every case.
Counter units and the gfx950 FETCH_SIZE correction: /opt/skills/guides/MI355X_MICROARCH.md, "HBM".  The x2
correction is calibrated for 16-byte-per-lane streams only, so every case carries its access width and two
calibration cases (a pure 16 B/lane stream and a pure 4 B/lane stream of known size) come first."""
import csv
import glob
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
REPS = 5


def cases(pkg):
    P = pkg.Plan
    ident = lambda l: np.arange(l, dtype=np.uint32)  # noqa: E731
    sel = lambda l: np.arange(l, dtype=np.int32)  # noqa: E731
    odd = [3653, 101, 271]
    n_odd = int(np.prod(odd))
    out = []

    def add(name, plan, n_in, n_out, n_read=None, width="16", dtype="float32"):
        elem = 8 if dtype == "float64" else 4
        out.append(dict(name=name, plan=plan, n_in=n_in, n_out=n_out, alg_bytes=((n_in if n_read is None else n_read) + n_out) * elem, width=width, dtype=dtype))

    # calibration: known bytes, one access width each
    add("calib rows 16B/lane [10]^8 dim0->all", P.drillup("float32", 0.0, "sum", [10] * 8, [1] + [10] * 7, [np.zeros(10, np.uint32)] + [ident(10)] * 7), 10 ** 8, 10 ** 7)
    add("calib rows 4B/lane [10,3001,3333] dim0->all", P.drillup("float32", 0.0, "sum", [10, 3001, 3333], [1, 3001, 3333], [np.zeros(10, np.uint32), ident(3001), ident(3333)]),
        10 * 3001 * 3333, 3001 * 3333, width="4")
    # the transposes
    add("reorder odd (product,location,day)", P.reorder("float32", 0.0, odd, [2, 1, 0]), n_odd, n_odd, width="4")
    add("reorder [10]^8 reversed", P.reorder("float32", 0.0, [10] * 8, list(range(7, -1, -1))), 10 ** 8, 10 ** 8)
    add("reorder 2-D [10^4,10^4]", P.reorder("float32", 0.0, [10000, 10000], [1, 0]), 10 ** 8, 10 ** 8)
    add("reorder C5 (sku,location,day)", P.reorder("float32", 0.0, [3652, 100, 274], [2, 1, 0]), 3652 * 100 * 274, 3652 * 100 * 274)
    # item 7: dice on odd extents
    s1 = [sel(3653), np.arange(0, 101, 3, dtype=np.int32), sel(271)]
    add("dice 34 of 101 locations (odd)", P.dice("float32", 0.0, odd, [3653, 34, 271], s1), n_odd, 3653 * 34 * 271, 3653 * 34 * 271, width="16/4")
    s0 = [np.arange(0, 3653, 3, dtype=np.int32), sel(101), sel(271)]
    add("dice every third day (odd)", P.dice("float32", 0.0, odd, [len(s0[0]), 101, 271], s0), n_odd, len(s0[0]) * 101 * 271, len(s0[0]) * 101 * 271, width="16/4")
    # item 7: drillUp regimes
    add("drillUp [1e5,1000]->10 interleaved (tile)", P.drillup("float32", 0.0, "sum", [100000, 1000], [100000, 10], [ident(100000), (np.arange(1000) % 10).astype(np.uint32)]), 10 ** 8, 10 ** 6)
    add("drillUp [1000,1000,100] interleaved (flat)", P.drillup("float32", 0.0, "sum", [1000, 1000, 100], [1000, 10, 100], [ident(1000), (np.arange(1000) % 10).astype(np.uint32), ident(100)]), 10 ** 8, 10 ** 6)
    add("drillUp [10]^8 dim0->all highest", P.drillup("float32", 0.0, "highest", [10] * 8, [1] + [10] * 7, [np.zeros(10, np.uint32)] + [ident(10)] * 7), 10 ** 8, 10 ** 7)
    add("drillUp [1e6,100]->[1,100] (reduce)", P.drillup("float32", 0.0, "sum", [10 ** 6, 100], [1, 100], [np.zeros(10 ** 6, np.uint32), ident(100)]), 10 ** 8, 100)
    add("drillUp [1e4,1e4]->[1,1e4] (row kernel over segments + fold)", P.drillup("float32", 0.0, "sum", [10 ** 4, 10 ** 4], [1, 10 ** 4], [np.zeros(10 ** 4, np.uint32), ident(10 ** 4)]), 10 ** 8, 10 ** 4)
    add("drillUp odd location->10 interleaved", P.drillup("float32", 0.0, "sum", odd, [3653, 10, 271], [ident(3653), (np.arange(101) % 10).astype(np.uint32), ident(271)]), n_odd, 3653 * 10 * 271, width="4")
    # round 3: the regimes that were below 0.70 and the load kernels
    add("drillUp [4e5,250]->[1,250] (reduce4, wide rows)", P.drillup("float32", 0.0, "sum", [400000, 250], [1, 250], [np.zeros(400000, np.uint32), ident(250)]), 10 ** 8, 250)
    add("drillUp [1e5,1000]->[1,1000] (reduce4, wide rows)", P.drillup("float32", 0.0, "sum", [10 ** 5, 1000], [1, 1000], [np.zeros(10 ** 5, np.uint32), ident(1000)]), 10 ** 8, 1000)
    add("drillUp [3001,3333,10]->11 groups (flat)", P.drillup("float32", 0.0, "sum", [3001, 3333, 10], [3001, 11, 10], [ident(3001), (np.arange(3333) // 303).astype(np.uint32), ident(10)]),
        3001 * 3333 * 10, 3001 * 11 * 10, width="4")
    add("drillUp C5 city->country", P.drillup("float32", 0.0, "sum", [3652, 100, 274], [3652, 10, 274], [ident(3652), (np.arange(100) // 10).astype(np.uint32), ident(274)]),
        3652 * 100 * 274, 3652 * 10 * 274, width="16")
    add("drillDown [12000,8192]->[120000,8192]", P.drilldown("float32", 0.0, "sum", [1200, 8192], [12000, 8192], [np.repeat(np.arange(1200), 10).astype(np.uint32), ident(8192)]),
        1200 * 8192, 12000 * 8192)
    i8 = [sel(10)] * 8
    add("load [10]^8 identity item maps", P.load("float32", 0.0, 0.0, [10] * 8, [10] * 8, i8), 10 ** 8, 10 ** 8)
    p4 = list(i8)
    p4[4] = np.array([3, 1, 4, 0, 9, 2, 6, 5, 8, 7], np.int32)
    add("load [10]^8 items of dim4 remapped", P.load("float32", 0.0, 0.0, [10] * 8, [10] * 8, p4), 10 ** 8, 10 ** 8)
    p7 = list(i8)
    p7[7] = np.array([3, 1, 4, 0, 9, 2, 6, 5, 8, 7], np.int32)
    add("load [10]^8 items of dim7 permuted (rows through LDS)", P.load("float32", 0.0, 0.0, [10] * 8, [10] * 8, p7), 10 ** 8, 10 ** 8)
    # Float64 cells (Float64 measures; integer measures of the Node host): the forms generalised for them in round 3
    add("Float64 drillUp [10]^8 dim0->all", P.drillup("float64", 0.0, "sum", [10] * 8, [1] + [10] * 7, [np.zeros(10, np.uint32)] + [ident(10)] * 7), 10 ** 8, 10 ** 7, dtype="float64")
    add("Float64 drillUp [1e8]->[1] (reduce4, two cells per lane)", P.drillup("float64", 0.0, "sum", [10 ** 8], [1], [np.zeros(10 ** 8, np.uint32)]), 10 ** 8, 1, dtype="float64")
    month = (np.arange(3652) // 30.4375).astype(np.uint32)
    add("Float64 drillUp [27400,3652] day innermost -> month (gtile)", P.drillup("float64", 0.0, "sum", [27400, 3652], [27400, int(month.max()) + 1], [ident(27400), month]),
        27400 * 3652, 27400 * (int(month.max()) + 1), dtype="float64")
    add("Float64 reorder [10]^8 reversed", P.reorder("float64", 0.0, [10] * 8, list(range(7, -1, -1))), 10 ** 8, 10 ** 8, dtype="float64")
    add("Float64 reorder 2-D [10^4,10^4]", P.reorder("float64", 0.0, [10000, 10000], [1, 0]), 10 ** 8, 10 ** 8, dtype="float64")
    return out


def probe(quiet):
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    from olap_in_memory_amd.sharded import HipEngine

    eng = HipEngine("cuda:0")
    L = pkg.lib()
    marker = eng.empty(4096, "float32")
    scratch = eng.empty(2048, "float32")
    meta = []
    for c in cases(pkg):
        vals = eng.empty(c["n_in"], c["dtype"])
        pkg.capi.check(L.olap_fill_seeded(vals.data_ptr(), None, c["n_in"], 0, 3 if c["dtype"] == "float64" else 2, 99, 1.0, eng.stream()))
        dst = eng.empty(c["n_out"], c["dtype"])
        args = (vals.data_ptr(), None, dst.data_ptr(), None, eng.stream())
        torch.cuda.synchronize()
        pkg.capi.check(L.olap_diag_read_ceiling(marker.data_ptr(), 4096 * 4, scratch.data_ptr(), eng.stream()))
        c["plan"].run(*args)  # warm-up: same kernels, inside the case's marker segment (medians are taken)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            c["plan"].run(*args)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / REPS * 1e3
        gbs = c["alg_bytes"] / (us * 1e-6) / 1e9
        meta.append({"name": c["name"], "kernel": c["plan"].kernel_name, "reps": REPS, "alg_bytes": c["alg_bytes"], "width": c["width"],
                     "us": us, "GBps": gbs, "frac": gbs / 8000})
        if not quiet:
            print("%-46s %9.1f us %9.1f GB/s  %.3f  %s" % (c["name"], us, gbs, gbs / 8000, c["plan"].kernel_name), flush=True)
        del vals, dst
        torch.cuda.empty_cache()
    pkg.capi.check(L.olap_diag_read_ceiling(marker.data_ptr(), 4096 * 4, scratch.data_ptr(), eng.stream()))
    torch.cuda.synchronize()
    os.makedirs(OUT, exist_ok=True)
    if not quiet:  # the un-profiled run owns the timings
        json.dump(meta, open(os.path.join(OUT, "pmc_cases.json"), "w"), indent=1)


def counters_by_case(dirname, names):
    """-> list (one per case) of {counter: median per-launch value summed over the case's kernels}"""
    files = glob.glob(os.path.join(OUT, dirname, "*", "*_counter_collection.csv"))
    if not files:
        return None
    rows = sorted(csv.DictReader(open(max(files, key=os.path.getmtime))), key=lambda r: (int(r["Start_Timestamp"]), int(r["Dispatch_Id"])))
    per_case, cur = [], None
    seen = set()
    for r in rows:
        if "diag_read_kernel" in r["Kernel_Name"]:
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                cur = {}
                per_case.append(cur)
            continue
        if cur is None or r["Counter_Name"] not in names or "fill_seeded" in r["Kernel_Name"]:
            continue
        cur.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    out = []
    for c in per_case[:-1] if len(per_case) > 1 else per_case:
        out.append({name: sum(statistics.median(v) for v in kernels.values()) for name, kernels in c.items()})
    return out


def summarize(tag):
    meta = json.load(open(os.path.join(OUT, "pmc_cases.json")))
    fetch = counters_by_case("pmc_fetch", {"FETCH_SIZE"})
    write = counters_by_case("pmc_write", {"WRITE_SIZE"})
    lds = counters_by_case("pmc_lds", {"SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"})
    tcc_names = {"TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_BUSY_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum",
                 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_TAG_STALL_sum"}
    tcc = [counters_by_case(d, tcc_names) for d in ("pmc_tcc_wr", "pmc_tcc_rd", "pmc_tcc_stall")]
    tlb_names = {"TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_STALL_MULTI_MISS_sum",
                 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum", "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum", "TCP_UTCL1_SERIALIZATION_STALL_sum", "TCP_UTCL1_THRASHING_STALL_sum",
                 "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_TA_ADDR_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_BUSY_sum"}
    tcc += [counters_by_case(d, tlb_names) for d in ("pmc_tlb", "pmc_tlb_stall", "pmc_tcp")]
    doc = {"note": "per launch; FETCH_SIZE / WRITE_SIZE in KiB (separate --pmc passes).  gfx950: FETCH_SIZE reports half the bytes of a 16 B/lane "
                   "stream (MI355X_MICROARCH.md); the two calibration cases give the factor for each access width on this box "
                   "(fetch_factor = algorithmic read bytes / (FETCH_SIZE * 1024)), and the factor of a case's width is applied to it.",
           "cases": []}
    factor = {}
    for i, m in enumerate(meta):
        c = dict(m)
        f = fetch[i].get("FETCH_SIZE") if fetch and i < len(fetch) else None
        w = write[i].get("WRITE_SIZE") if write and i < len(write) else None
        c["FETCH_SIZE_KiB"], c["WRITE_SIZE_KiB"] = f, w
        if lds and i < len(lds):
            c["SQ_LDS_BANK_CONFLICT"] = lds[i].get("SQ_LDS_BANK_CONFLICT")
            c["SQ_LDS_IDX_ACTIVE"] = lds[i].get("SQ_LDS_IDX_ACTIVE")
            if c["SQ_LDS_IDX_ACTIVE"]:
                c["lds_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        for t in tcc:  # memory-side (EA) requests of the L2 and its stall cycles, summed over the channels
            if t and i < len(t):
                c.update({k.replace("_sum", ""): v for k, v in t[i].items()})
        if c.get("TCC_EA0_WRREQ") and c.get("TCC_EA0_WRREQ_64B") is not None:
            c["partial_write_share"] = 1.0 - c["TCC_EA0_WRREQ_64B"] / c["TCC_EA0_WRREQ"]  # writes that are not whole 64-byte requests
        if c.get("TCC_BUSY"):
            for k in ("TCC_EA0_WRREQ_STALL", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL", "TCC_TOO_MANY_EA_WRREQS_STALL", "TCC_TAG_STALL"):
                if c.get(k) is not None:
                    c[k.lower() + "_per_busy"] = c[k] / c["TCC_BUSY"]
        if c.get("TCP_UTCL1_REQUEST"):
            c["tlb_miss_share"] = (c.get("TCP_UTCL1_TRANSLATION_MISS") or 0.0) / c["TCP_UTCL1_REQUEST"]
        if c.get("TCP_GATE_EN1"):
            for k in ("TCP_PENDING_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES", "TCP_TA_ADDR_STALL_CYCLES"):
                if c.get(k) is not None:
                    c[k.lower() + "_per_tcp_cycle"] = c[k] / c["TCP_GATE_EN1"]
        if m["name"].startswith("calib") and f and w is not None:
            read_bytes = m["alg_bytes"] - w * 1024  # what is left of the algorithmic bytes after the (exact) writes
            factor[m["width"]] = read_bytes / (f * 1024)
            c["fetch_factor"] = factor[m["width"]]
        doc["cases"].append(c)
    for c in doc["cases"]:
        f, w = c.get("FETCH_SIZE_KiB"), c.get("WRITE_SIZE_KiB")
        if f is None or w is None:
            continue
        k = factor.get("4" if c["width"] == "4" else "16", 2.0)
        c["hbm_bytes"] = k * f * 1024 + w * 1024
        c["traffic_over_algorithmic"] = c["hbm_bytes"] / c["alg_bytes"]
    doc["fetch_factor_by_width"] = factor
    # always "<tag>_kernels": profiles/traffic_<tag>.json is the headline launch's file that bench.py reads
    if not tag.endswith("_kernels"):
        tag += "_kernels"
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % tag)
    # keep the headline launch's figure where bench.py looks for it
    head = next((c for c in doc["cases"] if c["name"].startswith("calib rows 16B")), None)
    doc["hbm_bytes_per_launch"] = head.get("hbm_bytes") if head else None
    json.dump(doc, open(path, "w"), indent=1)
    # on the GPU box only gpurun_out/ travels back
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(doc, open(os.path.join(ROOT, "gpurun_out", os.path.basename(path)), "w"), indent=1)
    for c in doc["cases"]:
        print("%-50s %8.1f us  frac %.3f  traffic/alg %s  lds conflict share %s  partial writes %s  wrreq stall/busy %s  rd credit stall/busy %s" % (
            c["name"], c["us"], c["frac"], "%.2f" % c["traffic_over_algorithmic"] if "traffic_over_algorithmic" in c else "-",
            "%.2f" % c["lds_conflict_share"] if "lds_conflict_share" in c else "-", "%.2f" % c["partial_write_share"] if "partial_write_share" in c else "-",
            "%.2f" % c["tcc_ea0_wrreq_stall_per_busy"] if "tcc_ea0_wrreq_stall_per_busy" in c else "-",
            "%.2f" % c["tcc_ea0_rdreq_dram_credit_stall_per_busy"] if "tcc_ea0_rdreq_dram_credit_stall_per_busy" in c else "-")
            + ("  tlb miss share %.4f" % c["tlb_miss_share"] if "tlb_miss_share" in c else "")
            + ("  L1: pending-stall %.2f  L2-return-stall %.2f of its cycles" % (c.get("tcp_pending_stall_cycles_per_tcp_cycle", float("nan")), c.get("tcp_tcr_tcp_stall_cycles_per_tcp_cycle", float("nan")))
               if "tcp_pending_stall_cycles_per_tcp_cycle" in c else ""))
    print("wrote", path)


if __name__ == "__main__":
    if "--summarize" in sys.argv:
        summarize(sys.argv[sys.argv.index("--summarize") + 1])
    else:
        probe("--quiet" in sys.argv)
