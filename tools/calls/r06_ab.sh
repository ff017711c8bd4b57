#!/bin/bash
# round 6, call ab: the GPU tier serially in ONE process, as the driver runs it (after kMaxEntries 256 -> 4096), then under -n 4; the driver's bench line
O=gpurun_out/r06_ab; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -4 $O/suite_serial.log | tee -a $O/summary.txt
timeout 900 python3 -m pytest tests -q -m gpu -n 4 -rf --tb=long -p no:cacheprovider > $O/suite_n4.log 2>&1; tail -3 $O/suite_n4.log | tee -a $O/summary.txt; grep -n "^FAILED\|^ERROR" $O/suite_n4.log | head
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 1800 $O/bench_driver.json | tee -a $O/summary.txt
# (third run of this call: after the exit hook's bounded wait — the multi-device program beside the audit sweep again, 12 repetitions x 3 instances, as r06_final.sh does)
fails=0
for rep in $(seq 1 12); do
  pids=()
  for p in 0 1 2; do ( timeout 300 ./tests/cpp/test_multi_device 4 36 > $O/md_${rep}_$p.out 2> $O/md_${rep}_$p.err; echo $? > $O/md_${rep}_$p.rc ) & pids+=($!); done
  ( timeout 300 python3 -m pytest tests/test_gpu_pass1_sweep.py -q -m gpu -x -p no:cacheprovider > /dev/null 2>&1 ) & pids+=($!)
  wait "${pids[@]}"
  for p in 0 1 2; do rc=$(cat $O/md_${rep}_$p.rc); if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "multi-device FAIL rep $rep proc $p rc $rc"; tail -6 $O/md_${rep}_$p.err; fi; rm -f $O/md_${rep}_$p.rc $O/md_${rep}_$p.out $O/md_${rep}_$p.err; done
done
echo "multi-device beside the audit sweep, bounded exit wait: 12 repetitions x 3 instances: $fails failing" | tee -a $O/summary.txt
