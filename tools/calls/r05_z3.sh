# where the in-kernel checksum's time goes: 32-bit per-lane sums; ablations (wrong sums by design): hot sites off, cold sites off, both off
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle')[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec c5_in_kernel --no-cpu-baseline --c5 --frames 2000
GFW_JIT_DEFS="GFW_CK_ABLATE_HOT=1" rec c5_no_hot --no-cpu-baseline --c5 --frames 2000
GFW_JIT_DEFS="GFW_CK_ABLATE_COLD=1" rec c5_no_cold --no-cpu-baseline --c5 --frames 2000
GFW_JIT_DEFS="GFW_CK_ABLATE_HOT=1;GFW_CK_ABLATE_COLD=1" rec c5_neither --no-cpu-baseline --c5 --frames 2000
rec c5_pass --no-cpu-baseline --c5 --frames 2000 --sum-pass
