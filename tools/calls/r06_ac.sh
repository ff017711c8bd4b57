#!/bin/bash
# round 6, call ac: the single-frame GPU hunt on the final library (EWA included: the paired chroma launches), seeds 40000..43999, four shards side by side
O=gpurun_out/r06_ac; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
for s in 0 1 2 3; do
  a=$((40000 + s * 1000)); b=$((a + 1000))
  GFW_JIT_CACHE=/tmp/jitc$s timeout 2400 python3 tools/hunts/gpu_frame.py $a $b > $O/hunt_$s.log 2>&1 &
done
wait
for s in 0 1 2 3; do grep -v "amdgpu.ids\|^\.\.\. " $O/hunt_$s.log | tail -8 | tee -a $O/summary.txt; done
