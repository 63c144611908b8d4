set -o pipefail
timeout -k 10 600 python -m pytest tests/test_sharded_gloo.py tests/test_js_host.py -m gpu -x -q > gpurun_out/sharded_r03b.log 2>&1 || { tail -40 gpurun_out/sharded_r03b.log; exit 1; }
tail -3 gpurun_out/sharded_r03b.log
{
echo "== {0,0}, 10^7 cells"; OLAP_BENCH_DEVICES=0,0 timeout -k 10 120 node tools/js_sharded_bench.js
echo "== {0 x 8}, 10^7 cells"; OLAP_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout -k 10 120 node tools/js_sharded_bench.js
echo "== {0 x 8}, 10^9 cells"; OLAP_BENCH_DEVICES=0,0,0,0,0,0,0,0 OLAP_BENCH_SHAPE=320,5,5,5,5,5,5,10,20 timeout -k 10 200 node tools/js_sharded_bench.js
echo "== {0 x 8}, 10^9 cells, one issuing thread per rank (OLAP_SHARD_THREADS=1)"; OLAP_SHARD_THREADS=1 OLAP_BENCH_DEVICES=0,0,0,0,0,0,0,0 OLAP_BENCH_SHAPE=320,5,5,5,5,5,5,10,20 timeout -k 10 200 node tools/js_sharded_bench.js
echo "== {0 x 8}, 10^9 cells, round-2 form: unfused, issued by the calling thread (OLAP_SHARD_NO_FUSED=1)"; OLAP_SHARD_NO_FUSED=1 OLAP_BENCH_DEVICES=0,0,0,0,0,0,0,0 OLAP_BENCH_SHAPE=320,5,5,5,5,5,5,10,20 timeout -k 10 200 node tools/js_sharded_bench.js
} > gpurun_out/js_sharded_bench_r03.txt 2>&1
cat gpurun_out/js_sharded_bench_r03.txt
