# the private-segment lead: the checksum build of C2's kernel at 8 waves (4 dwords of scratch) and at 7 (none) under the occupancy counters
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
CMD="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --c5 --frames 400"
cd /tmp
GFW_JIT_WAVES=8 timeout 100 rocprofv3 -f csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/$O/w8 -o w8 -- $CMD > $GRAFT_REPO_ROOT/$O/w8.log 2>&1
timeout 100 rocprofv3 -f csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/$O/w7 -o w7 -- $CMD > $GRAFT_REPO_ROOT/$O/w7.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee -a $O/summary.txt
import csv, glob, os, collections
O = os.environ['O']
for tag in ('w8', 'w7'):
    acc = collections.defaultdict(list)
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (O, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('gfw_jit_kernel'):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    print(tag, {k: '%.4g' % v for k, v in sorted(m.items())}, 'n', len(next(iter(acc.values()), [])))
    if 'SQ_WAVE_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        print(tag, 'resident waves per SIMD (SQ_WAVE_CYCLES x 4 / GRBM_GUI_ACTIVE / 1024 SIMDs): %.2f' % (m['SQ_WAVE_CYCLES'] * 4 / m['GRBM_GUI_ACTIVE'] / 1024))
PY
rm -rf $O/w8 $O/w7
