# gfw_set_frame_checksums: the GPU tests, then C5 with the checksum taken in the warp kernel's store path against the gfw_checksum64 pass, and the driver command
timeout 600 python3 -m pytest tests/test_gpu_checksum.py -q -x 2>&1 | tail -15 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err; }
rec c5_2000_in_kernel --no-cpu-baseline --c5 --frames 2000
rec c5_2000_pass --no-cpu-baseline --c5 --frames 2000 --sum-pass
GFW_JIT_WAVES=7 rec c5_2000_in_kernel_w7 --no-cpu-baseline --c5 --frames 2000
rec driver --steps 20 --warmup 5
rec c2_200 --no-cpu-baseline --steps 200
