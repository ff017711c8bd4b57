#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pass1.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_matrix_builder.py -m gpu -x -q -s > gpurun_out/r02c/pytest.log 2>&1
echo "pytest rc=$?" | tee gpurun_out/r02c/summary.txt
tail -25 gpurun_out/r02c/pytest.log
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
echo "bench rc=$?" | tee -a gpurun_out/r02c/summary.txt
cat gpurun_out/r02c/bench.json; tail -3 gpurun_out/r02c/bench.err
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --variant 4 > gpurun_out/r02c/bench_v4.json 2> gpurun_out/r02c/bench_v4.err
cat gpurun_out/r02c/bench_v4.json
