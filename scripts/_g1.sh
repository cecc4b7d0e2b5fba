set -e
mkdir -p gpurun_out/r4k
timeout -k 10 500 python bench.py --no-cpu-baseline --no-configs --no-weak-tile --no-temporal-blocking > gpurun_out/r4k/bench_sw.json 2> gpurun_out/r4k/bench_sw.err
echo bench done
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/r4k/gpu_suite.log 2>&1
tail -3 gpurun_out/r4k/gpu_suite.log
