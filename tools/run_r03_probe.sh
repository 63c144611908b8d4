set -o pipefail
timeout -k 10 300 python -m pytest tests/test_totals.py tests/test_gpu_parity.py -m gpu -x -q -k "totals or load" > gpurun_out/totals_tests_r03.log 2>&1 || { tail -40 gpurun_out/totals_tests_r03.log; exit 1; }
tail -2 gpurun_out/totals_tests_r03.log
{ echo "== fused groups (round 3)"; timeout -k 10 200 python tools/totals_probe.py; echo "== OLAP_TOTALS_NO_GROUPS=1 (round 2)"; OLAP_TOTALS_NO_GROUPS=1 timeout -k 10 200 python tools/totals_probe.py; } > gpurun_out/totals_probe_r03.txt 2>&1
cat gpurun_out/totals_probe_r03.txt
timeout -k 10 200 python tools/sweep.py --only-load > gpurun_out/load_sweep_r03.txt 2>&1; cat gpurun_out/load_sweep_r03.txt
timeout -k 10 300 ./tools/pattern_ceiling.bin > gpurun_out/pattern_ceiling_r03.txt 2>&1
timeout -k 10 200 python tools/transpose_probe.py > gpurun_out/transpose_probe_r03.txt 2>&1; cat gpurun_out/transpose_probe_r03.txt
