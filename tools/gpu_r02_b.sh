#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee gpurun_out/r02b/summary.txt
tail -8 gpurun_out/r02b/pytest_gpu.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02b/bench_driver.json 2> gpurun_out/r02b/bench_driver.err
echo "bench driver rc=$?" | tee -a gpurun_out/r02b/summary.txt
timeout 600 python3 bench.py --c5 --no-cpu-baseline > gpurun_out/r02b/bench_c5.json 2> gpurun_out/r02b/bench_c5.err
echo "bench c5 rc=$?" | tee -a gpurun_out/r02b/summary.txt
cat gpurun_out/r02b/bench_c5.json
timeout 900 bash tools/profile.sh r02a > gpurun_out/r02b/profile.log 2>&1
echo "profile rc=$?" | tee -a gpurun_out/r02b/summary.txt
tail -60 gpurun_out/prof_r02a/summary.txt
