#!/bin/bash
# round 6, call u: the per-launch byte cap (clip_launch_limit): C3 through 16-frame calls, C4 capped and uncapped, C2 unchanged; the GPU tests that drive clip launches
O=gpurun_out/r06_u; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
C3="--no-cpu-baseline --width 7680 --height 4320 --resident 16 --steps 96"
rec c3_call16 $C3 --clip 16
GFW_CLIP_LAUNCH_MB=600 rec c3_call16_mb600 $C3 --clip 16 --no-parity
GFW_CLIP_LAUNCH_MB=2200 rec c3_call16_mb2200 $C3 --clip 16 --no-parity
rec c4_rgbaf --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 --steps 96
GFW_CLIP_LAUNCH_MB=8000 rec c4_rgbaf_uncapped --no-cpu-baseline --no-parity --fmt RGBAF32 --crop --resident 16 --steps 96
GFW_CLIP_LAUNCH_MB=2200 rec c4_rgbaf_mb2200 --no-cpu-baseline --no-parity --fmt RGBAF32 --crop --resident 16 --steps 96
rec c4_gbrapf32 --no-cpu-baseline --no-parity --fmt GBRAPF32LE --crop --resident 16 --steps 96
GFW_CLIP_LAUNCH_MB=8000 rec c4_gbrapf32_uncapped --no-cpu-baseline --no-parity --fmt GBRAPF32LE --crop --resident 16 --steps 96
rec rgbaf16 --no-cpu-baseline --no-parity --fmt RGBAF16 --resident 16 --steps 96
GFW_CLIP_LAUNCH_MB=8000 rec rgbaf16_uncapped --no-cpu-baseline --no-parity --fmt RGBAF16 --resident 16 --steps 96
rec c2_200 --no-cpu-baseline --steps 200
GFW_CLIP_LAUNCH_MB=600 rec c2_call16_mb600 --no-cpu-baseline --no-parity --steps 208 --clip 16
rec driver --steps 20 --warmup 5
timeout 1500 python3 -m pytest tests/test_gpu_jit.py tests/test_gpu_checksum.py tests/test_gpu_coalesce.py tests/test_gpu_fullsize.py tests/test_gpu_bench.py tests/test_gpu_multi_device.py -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee -a $O/summary.txt
