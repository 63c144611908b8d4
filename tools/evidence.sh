#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/evidence.sh'): GPU parity tests, smoke, the default bench line,
# and the three rocprofv3 passes the roofline object is checked against (kernel trace + stats, then
# FETCH_SIZE and WRITE_SIZE each in their own --pmc pass).  Everything lands in gpurun_out/;
# `python tools/summarize_profiles.py <tag>` condenses it into profiles/ afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
rm -rf "$O/prof_kt" "$O/prof_fetch" "$O/prof_write"
cd "$R" || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1 || { tail -20 "$O/pytest_gpu.log"; exit 1; }
tail -1 "$O/pytest_gpu.log"
timeout -k 10 120 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > "$O/smoke.log" 2>&1 || { tail -20 "$O/smoke.log"; exit 1; }
tail -1 "$O/smoke.log"
timeout -k 10 400 python bench.py > "$O/bench.json" 2> "$O/bench_err.log" || { tail -20 "$O/bench_err.log"; exit 1; }
cat "$O/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_kt" -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$O/prof_kt.log" 2>&1 || { tail -5 "$O/prof_kt.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/prof_fetch" -- python3 "$R/bench.py" --steps 20 --warmup 2 --no-cpu-baseline > "$O/prof_fetch.log" 2>&1 || { tail -5 "$O/prof_fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/prof_write" -- python3 "$R/bench.py" --steps 20 --warmup 2 --no-cpu-baseline > "$O/prof_write.log" 2>&1 || { tail -5 "$O/prof_write.log"; exit 1; }
cd "$R"
timeout -k 10 300 python tools/sweep.py > "$O/sweep.txt" 2>&1 && tail -3 "$O/sweep.txt"
timeout -k 10 300 python tools/sweep2.py > "$O/sweep_awkward.txt" 2>&1 && tail -3 "$O/sweep_awkward.txt"
DTYPE=float64 timeout -k 10 300 python tools/sweep.py 2>&1 | grep -v amdgpu.ids > "$O/sweep_f64.txt" && tail -3 "$O/sweep_f64.txt"
for d in float32 float64; do DTYPE=$d timeout -k 10 300 python tools/cross_sweep.py 2>&1 | grep -v amdgpu.ids; done > "$O/cross_sweep.txt" && tail -3 "$O/cross_sweep.txt"
timeout -k 10 200 python tools/dd_bench.py > "$O/dd_bench.txt" 2>&1 && tail -5 "$O/dd_bench.txt"
timeout -k 10 200 python tools/odd.py 2>&1 | grep -v amdgpu.ids > "$O/odd.txt" && cat "$O/odd.txt"
timeout -k 10 200 python tools/big.py 2>&1 | grep -v amdgpu.ids > "$O/big.txt" && cat "$O/big.txt"
timeout -k 10 200 node tools/js_bench.js > "$O/js_bench.txt" 2>&1 && tail -5 "$O/js_bench.txt"
timeout -k 10 200 node tools/js_reference_benchmark.js > "$O/js_reference_benchmark.txt" 2>&1 && tail -3 "$O/js_reference_benchmark.txt"
timeout -k 10 200 node tools/js_chain.js > "$O/js_chain.txt" 2>&1 && tail -3 "$O/js_chain.txt"
timeout -k 10 200 python tools/transpose_probe.py 2>&1 | grep -v amdgpu.ids > "$O/transpose_probe.txt" && tail -3 "$O/transpose_probe.txt"
if [ ! -x tools/tile_probe.bin ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOLAP_TILE_PROBE -I include -I olap-in-memory_amd/csrc tools/tile_probe.hip -o tools/tile_probe.bin; fi
timeout -k 10 120 ./tools/tile_probe.bin > "$O/tile_probe.txt" 2>&1 && tail -4 "$O/tile_probe.txt"
timeout -k 10 200 python tools/totals_probe.py 2>&1 | grep -v amdgpu.ids > "$O/totals_probe.txt" && tail -5 "$O/totals_probe.txt"
if [ ! -x tools/headline_limit.bin ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I olap-in-memory_amd/csrc tools/headline_limit.hip -o tools/headline_limit.bin; fi
timeout -k 10 120 ./tools/headline_limit.bin > "$O/headline_limit.txt" 2>&1 && tail -3 "$O/headline_limit.txt"
if [ ! -x tools/pattern_ceiling.bin ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pattern_ceiling.hip -o tools/pattern_ceiling.bin; fi
timeout -k 10 300 ./tools/pattern_ceiling.bin > "$O/pattern_ceiling.txt" 2>&1 && tail -3 "$O/pattern_ceiling.txt"
echo evidence done
# separately (each its own gpurun call): bash tools/pmc_run.sh (per-kernel PMC passes -> profiles/traffic_<tag>_kernels.json),
# bash tools/run_sharded_evidence.sh (sharded tests + tools/js_sharded_bench.js by device list and issuing mode)
