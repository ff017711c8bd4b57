#!/bin/bash
# round 6, call af: after the pairing rule was widened (any two single-channel planes without the colour-range fix): the single-frame hunt on 1 200 more seeds, four
# shards side by side; then the GPU tier serially in one process (the driver's way) and the driver's bench line
O=gpurun_out/r06_af; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
for s in 0 1 2 3; do
  a=$((60000 + s * 300)); b=$((a + 300))
  GFW_JIT_CACHE=/tmp/jitc$s timeout 1500 python3 tools/hunts/gpu_frame.py $a $b > $O/hunt_$s.log 2>&1 &
done
wait
for s in 0 1 2 3; do grep -v "amdgpu.ids\|^\.\.\. " $O/hunt_$s.log | tail -4 | tee -a $O/summary.txt; done
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -4 $O/suite_serial.log | tee -a $O/summary.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.json | tee -a $O/summary.txt
