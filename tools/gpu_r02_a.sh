#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench.py tests/test_gpu_pass1.py -m gpu -x -q > gpurun_out/r02a/pytest_new.log 2>&1
echo "pytest new rc=$?" | tee gpurun_out/r02a/summary.txt
tail -15 gpurun_out/r02a/pytest_new.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver.json 2> gpurun_out/r02a/bench_driver.err
echo "bench driver rc=$?" | tee -a gpurun_out/r02a/summary.txt
timeout 300 python3 bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
echo "bench default rc=$?" | tee -a gpurun_out/r02a/summary.txt
timeout 600 python3 bench.py --c5 --no-cpu-baseline > gpurun_out/r02a/bench_c5.json 2> gpurun_out/r02a/bench_c5.err
echo "bench c5 rc=$?" | tee -a gpurun_out/r02a/summary.txt
cat gpurun_out/r02a/bench_driver.json gpurun_out/r02a/bench_c5.json
tail -5 gpurun_out/r02a/*.err
