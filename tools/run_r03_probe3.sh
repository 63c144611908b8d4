set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reorder or transpose" > gpurun_out/reorder_tests_r03.log 2>&1 || { tail -40 gpurun_out/reorder_tests_r03.log; exit 1; }
tail -2 gpurun_out/reorder_tests_r03.log
{
for round in 1 2; do
echo "== one tile per workgroup (OLAP_XY_NO_STREAM=1), round $round"; OLAP_XY_NO_STREAM=1 timeout -k 10 200 python tools/transpose_probe.py 2>&1 | grep -v amdgpu.ids
echo "== streaming workgroups (next tile's loads in flight), round $round"; timeout -k 10 200 python tools/transpose_probe.py 2>&1 | grep -v amdgpu.ids
done
echo "== streaming, 3 workgroups per CU"; OLAP_XY_STREAM_WGS=3 timeout -k 10 200 python tools/transpose_probe.py 2>&1 | grep -v amdgpu.ids
echo "== streaming, 8 workgroups per CU asked (LDS holds 4)"; OLAP_XY_STREAM_WGS=8 timeout -k 10 200 python tools/transpose_probe.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/transpose_stream_ab_r03.txt 2>&1
cat gpurun_out/transpose_stream_ab_r03.txt
