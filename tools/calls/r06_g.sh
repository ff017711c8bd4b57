# the GoPro certified first pass after the trans-use hazard fix: diagnosis lines, the new test file, the jit / lens-model tests, the benches
timeout 300 python3 tools/diag_gopro.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 1200 python3 -m pytest tests/test_gpu_pass1_radial.py tests/test_gpu_jit.py tests/test_gpu_lens_models.py tests/test_gpu_multi_device.py -q -m gpu -x --tb=long -rA -p no:cacheprovider 2>&1 | grep -v "^PASSED" | tail -30 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec c2_gopro --no-cpu-baseline --lens-model gopro --steps 100
rec c2_gopro_lanczos --no-cpu-baseline --lens-model gopro --steps 100 --interp 8
rec c2_gopro_bicubic --no-cpu-baseline --lens-model gopro --steps 100 --interp 4
rec nv12_gopro --no-cpu-baseline --lens-model gopro --steps 100 --fmt NV12
