# r06 fifth call: a process that leaves main() while its first build compiles (old library against the one that loads the compiler up front); the GoPro certified
# first pass (audit builds, parity, bench); the whole suite
date
mkdir -p /tmp/oldlib && cp variants/libgfwarp_full_w5.so /tmp/oldlib/libgfwarp.so
for which in old new; do
  fails=0
  for rep in $(seq 1 25); do
    if [ $which = old ]; then LD_LIBRARY_PATH=/tmp/oldlib:$LD_LIBRARY_PATH GFW_JIT_CACHE= timeout 100 ./tests/cpp/test_multi_device exit > $O/exit_$which.out 2> $O/exit_$which.err; rc=$?
    else GFW_JIT_CACHE= timeout 100 ./tests/cpp/test_multi_device exit > $O/exit_$which.out 2> $O/exit_$which.err; rc=$?; fi
    if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "exit-during-build ($which library) rep $rep: rc $rc"; tail -3 $O/exit_$which.err; fi
  done
  echo "exit during the first build, $which library: 25 runs, $fails failing" | tee -a $O/summary.txt
done
date
timeout 1200 python3 -m pytest tests/test_gpu_pass1_radial.py tests/test_gpu_multi_device.py -q -m gpu -x --tb=long -rA -p no:cacheprovider 2>&1 | tail -40 | tee -a $O/summary.txt
date
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec c2_gopro --no-cpu-baseline --lens-model gopro --steps 100
rec c2_gopro_exact --no-cpu-baseline --lens-model gopro --steps 100 --variant 2
rec c2_gopro_lanczos --no-cpu-baseline --lens-model gopro --steps 100 --interp 8
rec c2_200 --no-cpu-baseline --steps 200
date
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -5 $O/suite_serial.log | tee -a $O/summary.txt
date
