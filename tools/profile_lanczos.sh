#!/bin/bash
# rocprofv3 counters for the Lanczos4 configuration of the bench (run on the GPU box)
set -u
OUT=gpurun_out/prof_l8
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --interp 8 --resident 8"
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/b1.log 2>&1
rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/b2.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
