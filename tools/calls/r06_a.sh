# r06 first call (verdict item 1): the certificate sweep beside other GPU processes, in a loop, with the assertion text kept; then the design probes
# (true cycles per instruction class, typed buffer loads) and HEAD's LUT-sampler numbers.
conc() {   # conc <tag> <reps> <env> : 4 concurrent pytest processes per repetition, each its own log; a failing process's full report is kept
  local tag=$1 reps=$2 envs="$3" fails=0
  for rep in $(seq 1 $reps); do
    pids=()
    for p in 0 1 2 3; do
      case "$tag:$p" in
        mixed:1) T="tests/test_gpu_fullsize.py" ;;
        mixed:2) T="tests/test_gpu_jit.py" ;;
        mixed:3) T="tests/test_gpu_parity.py" ;;
        *) T="tests/test_gpu_pass1_sweep.py" ;;
      esac
      ( env $envs GFW_PROC=$p timeout 600 python3 -m pytest $T -q -m gpu -x -rA --tb=long -p no:cacheprovider > $O/conc_${tag}_${rep}_$p.log 2>&1; echo $? > $O/conc_${tag}_${rep}_$p.rc ) &
      pids+=($!)
    done
    wait "${pids[@]}"
    for p in 0 1 2 3; do
      rc=$(cat $O/conc_${tag}_${rep}_$p.rc)
      if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "FAIL $tag rep $rep proc $p rc $rc"; cp $O/conc_${tag}_${rep}_$p.log $O/FAILED_${tag}_${rep}_$p.log; else rm -f $O/conc_${tag}_${rep}_$p.log; fi
      rm -f $O/conc_${tag}_${rep}_$p.rc
    done
  done
  echo "conc $tag: $reps repetitions x 4 processes [$envs]: $fails failing processes" | tee -a $O/summary.txt
}
date
./tools/microbench_cycles > $O/microbench_cycles.txt 2>&1; tail -40 $O/microbench_cycles.txt
./tools/tbuffer_probe > $O/tbuffer_probe.txt 2>&1; cat $O/tbuffer_probe.txt
date
conc sweep4 12 "GFW_X=1"
conc mixed 8 "GFW_X=1"
date
# the r05 conditions themselves: the whole GPU suite under -n 4, twice (the failure was seen once in one such run)
for i in 1 2; do timeout 900 python3 -m pytest tests -q -m gpu -n 4 -rf --tb=long -p no:cacheprovider > $O/suite_n4_$i.log 2>&1; tail -3 $O/suite_n4_$i.log | tee -a $O/summary.txt; grep -n "^FAILED\|^ERROR" $O/suite_n4_$i.log | head; done
date
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec c2_200 --no-cpu-baseline --steps 200
rec c2_bicubic --no-cpu-baseline --interp 4
rec c2_lanczos --no-cpu-baseline --interp 8 --steps 100
rec nv12_lanczos --no-cpu-baseline --fmt NV12 --interp 8 --steps 100
rec c2_gopro --no-cpu-baseline --lens-model gopro --steps 100
rec host --no-cpu-baseline --host-buffers --steps 20
date
