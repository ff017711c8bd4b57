#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_pass1.py tests/test_gpu_lens_models.py tests/test_gpu_golden.py tests/test_golden.py -m gpu -x -q > gpurun_out/r02i/pytest.log 2>&1
echo "pytest rc=$?" | tee gpurun_out/r02i/summary.txt
tail -5 gpurun_out/r02i/pytest.log
for v in 0 5; do
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --variant $v > gpurun_out/r02i/bench_v$v.json 2> gpurun_out/r02i/bench_v$v.err
python3 -c "import json; d=json.load(open('gpurun_out/r02i/bench_v$v.json')); print('variant',$v,d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])"
done
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --c1 > gpurun_out/r02i/bench_c1.json 2> gpurun_out/r02i/bench_c1.err
python3 -c "import json; d=json.load(open('gpurun_out/r02i/bench_c1.json')); print('c1',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])"
