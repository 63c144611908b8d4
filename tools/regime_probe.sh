#!/bin/bash
# On the GPU box: per-kernel time split (rocprofv3 --kernel-trace --stats) of the cases given, one process each.
# usage: bash tools/regime_probe.sh <tag> case [case ...]     -> gpurun_out/regime_<tag>.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
TAG=$1; shift
mkdir -p "$O"; : > "$O/regime_$TAG.txt"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf "$O/regime_prof"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/regime_prof" -- python3 "$R/tools/regime_probe.py" $c > "$O/regime_run.log" 2>&1 || { tail -5 "$O/regime_run.log"; exit 1; }
  grep -v amdgpu.ids "$O/regime_run.log" | grep "us " >> "$O/regime_$TAG.txt"
  f=$(find "$O/regime_prof" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" >> "$O/regime_$TAG.txt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 30 and "fill" not in r["Name"]:
        print("    %-90s calls %4s  mean %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf "$O/regime_prof"
cat "$O/regime_$TAG.txt"
