# r06 third call: the whole GPU suite serially and under -n 4 on the tree with rects on the fused kernel, the checksum-ring fix, no host pinning;
# true cycles per instruction class (independent stages); the ahead-of-time kernels at 5 waves (no scratch) against 6
date
./tools/microbench_cycles > $O/microbench_cycles.txt 2>&1; cat $O/microbench_cycles.txt
date
timeout 1200 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -5 $O/suite_serial.log | tee -a $O/summary.txt
date
for i in 1 2; do timeout 900 python3 -m pytest tests -q -m gpu -n 4 -rf --tb=long -p no:cacheprovider > $O/suite_n4_$i.log 2>&1; tail -3 $O/suite_n4_$i.log | tee -a $O/summary.txt; grep -n "^FAILED\|^ERROR" $O/suite_n4_$i.log | head; done
date
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec driver --steps 20 --warmup 5
rec host --no-cpu-baseline --host-buffers --steps 40
rec c2_aot --no-cpu-baseline --steps 200 --jit 0 --clip 1
bash tools/gpu_ab.sh r06_c_ab "aot_w6:--jit 0 --clip 1" "aot_w5:--jit 0 --clip 1" "aot_w6:--jit 0 --clip 1" "aot_w5:--jit 0 --clip 1" 2>&1 | tail -6 | tee -a $O/summary.txt
date
