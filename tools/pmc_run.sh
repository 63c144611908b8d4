#!/bin/bash
# Runs on the GPU box: tools/pmc_probe.py un-profiled (timings), then one rocprofv3 --pmc pass per counter set.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
rm -rf "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_lds"
cd "$R" || exit 1
timeout -k 10 300 python3 tools/pmc_probe.py 2>&1 | grep -v amdgpu.ids | tee "$O/pmc_probe.txt" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_fetch.log" 2>&1 || { tail -5 "$O/pmc_fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_write.log" 2>&1 || { tail -5 "$O/pmc_write.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_lds" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_lds.log" 2>&1 || { tail -5 "$O/pmc_lds.log"; exit 1; }
# the L2's memory-side requests and stalls (TCC block: four counters a pass)
rm -rf "$O/pmc_tcc_wr" "$O/pmc_tcc_rd" "$O/pmc_tcc_stall"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum --output-format csv -d "$O/pmc_tcc_wr" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_tcc_wr.log" 2>&1 || { tail -5 "$O/pmc_tcc_wr.log"; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d "$O/pmc_tcc_rd" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_tcc_rd.log" 2>&1 || { tail -5 "$O/pmc_tcc_rd.log"; }
timeout -k 10 300 rocprofv3 --pmc TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum --output-format csv -d "$O/pmc_tcc_stall" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_tcc_stall.log" 2>&1 || { tail -5 "$O/pmc_tcc_stall.log"; }
# address translation and the vector L1 (TCP block)
rm -rf "$O/pmc_tlb" "$O/pmc_tlb_stall" "$O/pmc_tcp"
timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum --output-format csv -d "$O/pmc_tlb" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_tlb.log" 2>&1 || { tail -5 "$O/pmc_tlb.log"; }
timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum --output-format csv -d "$O/pmc_tlb_stall" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_tlb_stall.log" 2>&1 || { tail -5 "$O/pmc_tlb_stall.log"; }
timeout -k 10 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d "$O/pmc_tcp" -- python3 "$R/tools/pmc_probe.py" --quiet > "$O/pmc_tcp.log" 2>&1 || { tail -5 "$O/pmc_tcp.log"; }
cd "$R" && python3 tools/pmc_probe.py --summarize ${OLAP_PMC_TAG:-probe} | tee "$O/pmc_summary.txt"
