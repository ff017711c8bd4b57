# the lane-row's stores added up unshifted (one shift per lane at the fold): C5 again
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
timeout 300 python3 -m pytest tests/test_gpu_checksum.py -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
rec c5_2000_in_kernel --no-cpu-baseline --c5 --frames 2000
GFW_JIT_WAVES=7 rec c5_2000_in_kernel_w7 --no-cpu-baseline --c5 --frames 2000
rec c5_2000_pass --no-cpu-baseline --c5 --frames 2000 --sum-pass
