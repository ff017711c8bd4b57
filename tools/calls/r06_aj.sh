#!/bin/bash
# round 6, call aj: after the EWA loops' bounds fix (new kernel source id): the suites that reach the per-plane kernel, the EWA benches, rocprofv3 trace + PMC passes of
# the default workload with the traffic file re-stamped, the GPU tier serially in one process, the driver's line
O=gpurun_out/r06_aj; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_ewa_pair.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_fused_coverage.py -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
rec c2_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2
rec c1_ewa10 --no-cpu-baseline --interp 10 --steps 10 --warmup 2 --c1
rec yuv420p_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2 --fmt YUV420P
bash tools/profile_pmc.sh r06_final 2>&1 | grep -v "at::native" | head -60 > $O/profile.txt; grep -A3 "gfw_jit_kernel" $O/profile.txt | head -8 | tee -a $O/summary.txt
python3 tools/traffic_json.py gpurun_out/prof_r06_final $O/r06_c2_traffic.json 10
cp $O/r06_c2_traffic.json profiles/r06_c2_traffic.json
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -4 $O/suite_serial.log | tee -a $O/summary.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.json | tee -a $O/summary.txt
