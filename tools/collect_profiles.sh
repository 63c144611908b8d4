#!/bin/bash
# After `gpurun -- bash tools/evidence.sh`: condenses gpurun_out/ into profiles/ under the tag given (e.g. r03).
set -e
TAG=${1:?usage: tools/collect_profiles.sh <tag>}
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python tools/summarize_profiles.py "$TAG" > /dev/null
cd gpurun_out
for pair in bench.json:bench_$TAG.json sweep.txt:sweep_$TAG.txt sweep_awkward.txt:sweep_awkward_$TAG.txt sweep_f64.txt:sweep_f64_$TAG.txt cross_sweep.txt:cross_sweep_$TAG.txt \
            odd.txt:odd_extents_$TAG.txt dd_bench.txt:drilldown_$TAG.txt big.txt:config4_single_gpu_$TAG.txt js_bench.txt:js_bench_$TAG.txt js_chain.txt:js_chain_$TAG.txt \
            js_reference_benchmark.txt:js_reference_benchmark_$TAG.txt transpose_probe.txt:transpose_probe_$TAG.txt tile_probe.txt:tile_probe_$TAG.txt \
            totals_probe.txt:totals_$TAG.txt pattern_ceiling.txt:pattern_ceiling_$TAG.txt headline_limit.txt:headline_limit_$TAG.txt; do
  s=${pair%%:*}; d=${pair##*:}
  if [ -s "$s" ]; then grep -v "amdgpu.ids" "$s" > "../profiles/$d"; else echo "missing $s"; fi
done
