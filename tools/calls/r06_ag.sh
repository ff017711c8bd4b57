#!/bin/bash
# round 6, call ag: the multi-device / exit tests and the JIT tests on the library with the hardened exit hook; exit during the first build, 30 times beside the audit sweep
O=gpurun_out/r06_ag; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_multi_device.py tests/test_gpu_jit.py tests/test_gpu_jit_cache.py tests/test_gpu_threads.py -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/summary.txt
fails=0
for rep in $(seq 1 10); do
  pids=()
  for p in 0 1 2; do ( GFW_JIT_CACHE=/tmp/none_$rep_$p timeout 300 ./tests/cpp/test_multi_device exit > $O/ex_${rep}_$p.out 2> $O/ex_${rep}_$p.err; echo $? > $O/ex_${rep}_$p.rc ) & pids+=($!); done
  ( timeout 300 python3 -m pytest tests/test_gpu_pass1_sweep.py -q -m gpu -x -p no:cacheprovider > /dev/null 2>&1 ) & pids+=($!)
  wait "${pids[@]}"
  for p in 0 1 2; do rc=$(cat $O/ex_${rep}_$p.rc); if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "exit FAIL rep $rep proc $p rc $rc"; tail -4 $O/ex_${rep}_$p.err; fi; rm -f $O/ex_${rep}_$p.rc $O/ex_${rep}_$p.out $O/ex_${rep}_$p.err; done
done
echo "exit during the first build beside the audit sweep: 10 repetitions x 3 instances: $fails failing" | tee -a $O/summary.txt
