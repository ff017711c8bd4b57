#!/bin/bash
# round 6, call y (and z, after the row prefetch): EWA on planar chroma — U and V in one launch (gfw_plane_kernel<.., DUAL>): its test, the suites that reach the per-plane kernel, the EWA benches
O=gpurun_out/r06_z; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_ewa_pair.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_fused_coverage.py tests/test_gpu_coalesce.py tests/test_gpu_checksum.py tests/test_gpu_abi_errors.py -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -6 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
rec c2_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2
rec yuv420p_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2 --fmt YUV420P
rec c1_ewa10 --no-cpu-baseline --interp 10 --steps 10 --warmup 2 --c1
rec c2_1080p_ewa10_parity --no-cpu-baseline --interp 10 --steps 4 --warmup 1 --width 1920 --height 1080
