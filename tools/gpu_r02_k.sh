#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
for v in 0 5; do
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --variant $v > gpurun_out/r02k/bench_v$v.json 2> gpurun_out/r02k/bench_v$v.err
python3 -c "import json; d=json.load(open('gpurun_out/r02k/bench_v$v.json')); print('variant',$v,d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])" | tee -a gpurun_out/r02k/summary.txt
done
timeout 1200 python -m pytest tests/test_gpu_matrix_builder.py tests/test_gpu_ref_opencl.py tests/test_gpu_math.py tests/test_gpu_abi_errors.py tests/test_gpu_lens_models.py tests/test_gpu_parity.py tests/test_gpu_pass1.py tests/test_gpu_fullsize.py -m gpu -q -s > gpurun_out/r02k/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r02k/summary.txt
grep -E "passed|failed|Error|identical|certified second|FAILED" gpurun_out/r02k/pytest.log | tail -25
