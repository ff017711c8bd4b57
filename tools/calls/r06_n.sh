#!/bin/bash
# round 6, call n: the new clip-launch test over every lens model; the GPU clip hunt, four shards side by side
O=gpurun_out/r06_n; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_lens_models.py -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee -a $O/summary.txt
for s in 0 1 2 3; do
  a=$((5000 + s * 400)); b=$((a + 400))
  GFW_JIT_CACHE=/tmp/jitc$s timeout 1500 python3 tools/hunts/gpu_clip.py $a $b > $O/hunt_$s.log 2>&1 &
done
wait
for s in 0 1 2 3; do grep -v "amdgpu.ids\|^\.\.\. " $O/hunt_$s.log | tail -12 | tee -a $O/summary.txt; done
