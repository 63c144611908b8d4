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
cd "$R" && python3 tools/pmc_probe.py --summarize probe | tee "$O/pmc_summary.txt"
