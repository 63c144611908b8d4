set -o pipefail
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_js_host.py tests/test_bench_contract.py tests/test_sharded_gloo.py -m gpu -x -q > gpurun_out/tests_r03d.log 2>&1 || { tail -40 gpurun_out/tests_r03d.log; exit 1; }
tail -2 gpurun_out/tests_r03d.log
timeout -k 10 300 python tools/sweep2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep_awkward.txt; grep "3001,3333\|gtile" gpurun_out/sweep_awkward.txt
