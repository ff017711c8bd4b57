# the finisher split over 16 workgroups per frame: C5 again, plus a kernel trace of a short C5 run (which kernel takes what)
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
timeout 300 python3 -m pytest tests/test_gpu_checksum.py -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
rec c5_2000_in_kernel --no-cpu-baseline --c5 --frames 2000
rec c5_2000_pass --no-cpu-baseline --c5 --frames 2000 --sum-pass
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_c5 -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-parity --c5 --frames 800 > $GRAFT_REPO_ROOT/$O/trace_c5.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee -a $O/summary.txt
import csv, glob, os
for f in glob.glob(os.environ.get('O','gpurun_out/r05_z') + '/trace_c5/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
