cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or boundaries or awkward" > gpurun_out/tile.log 2>&1; tail -6 gpurun_out/tile.log
for d in float32 float64; do DTYPE=$d timeout -k 10 200 python3 tools/cross_sweep.py c5_product tile_interleaved rows1000_all groups11 2>&1 | grep -v amdgpu.ids; done
