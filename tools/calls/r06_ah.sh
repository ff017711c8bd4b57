#!/bin/bash
# round 6, call ah: the last library of the round — the GPU tier serially in one process (the driver's way), smoke, the driver's bench line
O=gpurun_out/r06_ah; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -4 $O/suite_serial.log | tee -a $O/summary.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 400 $O/bench_driver.json | tee -a $O/summary.txt
timeout 300 python3 bench.py --gpus 1 > $O/bench_default.json 2> $O/bench_default.err; python3 -c "import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['roofline']['frac'])" | tee -a $O/summary.txt
