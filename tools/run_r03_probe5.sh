set -o pipefail
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_insertion_order.py -m gpu -x -q > gpurun_out/tests_r03e.log 2>&1 || { tail -40 gpurun_out/tests_r03e.log; exit 1; }
tail -2 gpurun_out/tests_r03e.log
timeout -k 10 300 python tools/sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep.txt; cat gpurun_out/sweep.txt
