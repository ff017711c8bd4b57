# r06 fourth call: where does tests/cpp/test_multi_device stop under load (progress markers, watchdog, _exit on a failed check)?  rects after the alignment fix; the
# ahead-of-time kernels at five waves per SIMD (no scratch) against six
date
fails=0
for rep in $(seq 1 10); do
  pids=()
  for p in 0 1 2; do ( timeout 300 ./tests/cpp/test_multi_device 4 36 > $O/md_${rep}_$p.out 2> $O/md_${rep}_$p.err; echo $? > $O/md_${rep}_$p.rc ) & pids+=($!); done
  ( timeout 300 python3 -m pytest tests/test_gpu_pass1_sweep.py -q -m gpu -x -p no:cacheprovider > /dev/null 2>&1 ) & pids+=($!)
  wait "${pids[@]}"
  for p in 0 1 2; do rc=$(cat $O/md_${rep}_$p.rc); if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "multi-device FAIL rep $rep proc $p rc $rc"; tail -12 $O/md_${rep}_$p.err; else rm -f $O/md_${rep}_$p.out $O/md_${rep}_$p.err; fi; rm -f $O/md_${rep}_$p.rc; done
done
echo "multi-device under load: 10 repetitions x 3 instances beside the audit sweep: $fails failing" | tee -a $O/summary.txt
date
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -5 $O/suite_serial.log | tee -a $O/summary.txt
date
timeout 900 python3 -m pytest tests -q -m gpu -n 4 -rf --tb=long -p no:cacheprovider > $O/suite_n4_1.log 2>&1; tail -3 $O/suite_n4_1.log | tee -a $O/summary.txt; grep -n "^FAILED\|^ERROR" $O/suite_n4_1.log | head
date
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
for i in 1 2; do
rec c2_aot_w6_$i --no-cpu-baseline --steps 200 --jit 0 --clip 1
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_full_w5.so rec c2_aot_w5_$i --no-cpu-baseline --steps 200 --jit 0 --clip 1
done
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_full_w5.so rec nv12_aot_w5 --no-cpu-baseline --steps 200 --jit 0 --clip 1 --fmt NV12
rec nv12_aot_w6 --no-cpu-baseline --steps 200 --jit 0 --clip 1 --fmt NV12
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_full_w5.so rec lanczos_aot_w5 --no-cpu-baseline --steps 100 --jit 0 --clip 1 --interp 8
rec lanczos_aot_w6 --no-cpu-baseline --steps 100 --jit 0 --clip 1 --interp 8
date
