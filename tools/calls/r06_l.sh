#!/bin/bash
# round 6, call l: after the pointer-test fix — the radial tests (clip launches included), lens models, jit; A/B benches of the served models; then the whole GPU suite
O=gpurun_out/r06_l; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_pass1_radial.py tests/test_gpu_lens_models.py tests/test_gpu_jit.py -q -m gpu -x --tb=short -rA -s -p no:cacheprovider -n 4 2>&1 | grep -v "^PASSED\|amdgpu.ids" | tail -25 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; grep "differs from" $O/bench_$name.err | tee -a $O/summary.txt; }
for m in sony generic_polynomial gopro; do
  rec c2_$m --no-cpu-baseline --lens-model $m --steps 100
  GFW_P1_RADIAL=0 rec c2_${m}_exact --no-cpu-baseline --lens-model $m --steps 100
done
rec c2_gp_lanczos --no-cpu-baseline --lens-model generic_polynomial --steps 100 --interp 8
rec nv12_gp --no-cpu-baseline --lens-model generic_polynomial --steps 100 --fmt NV12
rec c2_poly5 --no-cpu-baseline --lens-model poly5 --steps 100
rec c2_gp_aot --no-cpu-baseline --lens-model generic_polynomial --steps 100 --jit 0
rec c2 --no-cpu-baseline --steps 200
timeout 2400 python3 -m pytest tests -q -m gpu -x --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee -a $O/summary.txt
