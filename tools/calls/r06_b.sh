# r06 second call: the audit-reset fix under the conditions that reproduced the failure (r06_a: 5 failing runs of ~90), the new tests, HOST buffers without the
# destination upload + page-locked caller ranges, the counter calibration, the driver line with 10 + 10 launches
conc() {
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
      ( env $envs timeout 600 python3 -m pytest $T -q -m gpu -x -rA --tb=long -p no:cacheprovider > $O/conc_${tag}_${rep}_$p.log 2>&1; echo $? > $O/conc_${tag}_${rep}_$p.rc ) &
      pids+=($!)
    done
    wait "${pids[@]}"
    for p in 0 1 2 3; do
      rc=$(cat $O/conc_${tag}_${rep}_$p.rc)
      if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "FAIL $tag rep $rep proc $p rc $rc"; cp $O/conc_${tag}_${rep}_$p.log $O/FAILED_${tag}_${rep}_$p.log; fi
      rm -f $O/conc_${tag}_${rep}_$p.log $O/conc_${tag}_${rep}_$p.rc
    done
  done
  echo "conc $tag: $reps repetitions x 4 processes [$envs]: $fails failing processes" | tee -a $O/summary.txt
}
date
timeout 900 python3 -m pytest tests/test_gpu_pass1_concurrent.py tests/test_gpu_multi_device.py tests/test_gpu_abi_errors.py tests/test_gpu_checksum.py tests/test_gpu_coalesce.py -q -m gpu -x --tb=long -p no:cacheprovider 2>&1 | tail -15 | tee -a $O/summary.txt
date
conc sweep4 14 "GFW_X=1"
conc mixed 8 "GFW_X=1"
date
for i in 1 2; do timeout 900 python3 -m pytest tests -q -m gpu -n 4 -rf --tb=long -p no:cacheprovider > $O/suite_n4_$i.log 2>&1; tail -3 $O/suite_n4_$i.log | tee -a $O/summary.txt; grep -n "^FAILED\|^ERROR" $O/suite_n4_$i.log | head; done
date
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec driver --steps 20 --warmup 5
rec c2_200 --no-cpu-baseline --steps 200
rec host --no-cpu-baseline --host-buffers --steps 40
GFW_PIN_HOST=0 rec host_nopin --no-cpu-baseline --host-buffers --steps 40
date
# counter calibration: FETCH_SIZE / WRITE_SIZE of kernels that move a known byte count once (tools/fetch_calib.hip), separate passes
export TMPDIR=/tmp
mkdir -p $O/calib
timeout 200 rocprofv3 -f csv --pmc FETCH_SIZE -d $O/calib/f -o f -- ./tools/fetch_calib > $O/calib/f.log 2>&1
timeout 200 rocprofv3 -f csv --pmc WRITE_SIZE -d $O/calib/w -o w -- ./tools/fetch_calib > $O/calib/w.log 2>&1
python3 - <<PY | tee -a $O/summary.txt
import csv, glob
for d in ("f", "w"):
    for f in glob.glob("$O/calib/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "calib_" in row["Kernel_Name"]:
                v = float(row["Counter_Value"])
                print("%-24s %-10s = %12.1f KiB  -> bytes moved / (counter x 1024) = %.3f" % (row["Kernel_Name"].split("(")[0], row["Counter_Name"], v, (1 << 30) / (v * 1024.0) if v else float("nan")))
PY
rm -rf $O/calib/f $O/calib/w
date
