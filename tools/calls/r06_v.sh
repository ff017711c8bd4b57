#!/bin/bash
# round 6, call v: the new launch-cap test; the GPU clip hunt with the cap at 1 MB (every call of 3-5 frames leaves in several launches), seeds 30000..33999, four shards
O=gpurun_out/r06_v; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_jit.py -q -m gpu --tb=short -p no:cacheprovider -k "capped or clip_entry" 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee -a $O/summary.txt
for s in 0 1 2 3; do
  a=$((30000 + s * 1000)); b=$((a + 1000))
  GFW_CLIP_LAUNCH_MB=1 GFW_JIT_CACHE=/tmp/jitc$s timeout 1500 python3 tools/hunts/gpu_clip.py $a $b > $O/hunt_$s.log 2>&1 &
done
wait
for s in 0 1 2 3; do grep -v "amdgpu.ids\|^\.\.\. " $O/hunt_$s.log | tail -6 | tee -a $O/summary.txt; done
