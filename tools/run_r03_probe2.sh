set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or reduce or few_outputs or batch or multi" > gpurun_out/reduce_tests_r03.log 2>&1 || { tail -40 gpurun_out/reduce_tests_r03.log; exit 1; }
tail -2 gpurun_out/reduce_tests_r03.log
timeout -k 10 300 python tools/sweep2.py > gpurun_out/sweep_awkward.txt 2>&1; grep -v amdgpu.ids gpurun_out/sweep_awkward.txt
