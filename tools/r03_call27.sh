#!/bin/bash
# round 3, twenty-seventh GPU call: libgfwarp against the fixture written by the reference's own kernel (tests/test_ref_golden.py -m gpu), the gfx950
# twin's explained-residual test under the extended classifier, the driver's bench command (without the CPU leg)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
timeout 110 python -m pytest tests/test_ref_golden.py tests/test_gpu_ref_opencl.py -m gpu -q -p no:cacheprovider > $O/ref_golden_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/ref_golden_gpu.log
timeout 50 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; d = json.load(open('$O/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['backend'], d['config']['parity_vs_oracle'])"
