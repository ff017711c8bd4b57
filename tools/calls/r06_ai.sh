#!/bin/bash
# round 6, call ai: the clip hunt on the last library of the round, seeds 70000..75999, four shards side by side (default launch cap)
O=gpurun_out/r06_ai; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
for s in 0 1 2 3; do
  a=$((70000 + s * 1500)); b=$((a + 1500))
  GFW_JIT_CACHE=/tmp/jitc$s timeout 2000 python3 tools/hunts/gpu_clip.py $a $b > $O/hunt_$s.log 2>&1 &
done
wait
for s in 0 1 2 3; do grep -v "amdgpu.ids\|^\.\.\. " $O/hunt_$s.log | tail -4 | tee -a $O/summary.txt; done
