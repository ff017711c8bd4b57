#!/bin/bash
# round 4, GPU call 15: the derived certificate (E from the kernel's own view of the mid-row matrix): first-pass audits, the GPU tier, the driver command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pass1.py tests/test_gpu_pass1_sweep.py -m gpu -q -p no:cacheprovider > $O/new_tests.log 2>&1; echo "pass1 tests rc $?" | tee -a $O/summary.txt; tail -12 $O/new_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_frame'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config']['parity_vs_oracle'])" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
