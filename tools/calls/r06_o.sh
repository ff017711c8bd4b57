#!/bin/bash
# round 6, call o: the GPU clip hunt at length — seeds 10000..17999, four shards side by side; then the single-frame generator (test_gpu_fuzz's) on 2000 more seeds
O=gpurun_out/r06_o; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
for s in 0 1 2 3; do
  a=$((10000 + s * 2000)); b=$((a + 2000))
  GFW_JIT_CACHE=/tmp/jitc$s timeout 2400 python3 tools/hunts/gpu_clip.py $a $b > $O/hunt_$s.log 2>&1 &
done
wait
for s in 0 1 2 3; do grep -v "amdgpu.ids\|^\.\.\. " $O/hunt_$s.log | tail -12 | tee -a $O/summary.txt; done
