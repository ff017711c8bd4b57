#!/bin/bash
# round 6, call h: the certified first pass of the closed-form radial models (sony, generic_polynomial, poly3, poly5, ptlens): audit tests, parity, A/B benches
O=gpurun_out/r06_h; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_pass1_radial.py tests/test_gpu_lens_models.py tests/test_gpu_jit.py -q -m gpu -x --tb=long -rA -s -p no:cacheprovider -n 4 2>&1 | grep -v "^PASSED\|amdgpu.ids" | tail -40 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
for m in sony generic_polynomial poly3 poly5 ptlens gopro; do
  rec c2_$m --no-cpu-baseline --lens-model $m --steps 100
  GFW_P1_RADIAL=0 rec c2_${m}_exact --no-cpu-baseline --lens-model $m --steps 100
done
rec c2_sony_lanczos --no-cpu-baseline --lens-model sony --steps 100 --interp 8
rec nv12_sony --no-cpu-baseline --lens-model sony --steps 100 --fmt NV12
rec c2 --no-cpu-baseline --steps 200
