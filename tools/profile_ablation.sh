#!/bin/bash
# tools/profile_ablation.sh "<variants>" — PMC counters of the fused kernel under timing ablations (GFW_OPT_KERNEL_VARIANT 16+bits)
export TMPDIR=/tmp
OUT=gpurun_out/prof_abl
rm -rf $OUT; mkdir -p $OUT
for v in ${1:-31 0}; do
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD -d $OUT/v$v -o p -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --variant $v > $OUT/log$v.txt 2>&1
rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH -d $OUT/w$v -o p -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --variant $v > $OUT/logw$v.txt 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/t$v -o t -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --variant $v > $OUT/logt$v.txt 2>&1
done
for d in $OUT/*; do if [ -d $d ]; then echo == $d; python3 - <<PY
import csv,glob
from collections import defaultdict
for f in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    acc=defaultdict(list)
    for row in csv.DictReader(open(f)):
        if 'gfw_' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k,v in sorted(acc.items()): print('   %-24s %.5g'%(k,sum(v)/len(v)))
for f in glob.glob("$d/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print('   ', row['Name'][:60], row['Calls'], row['AverageNs'])
PY
fi; done
