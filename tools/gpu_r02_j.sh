#!/bin/bash
cd $GRAFT_REPO_ROOT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 TMPDIR=/tmp
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --resident 16"
OUT=gpurun_out/r02j/prof; mkdir -p $OUT
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 -f csv --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
python3 tools/summarize_prof.py $OUT 2>&1 | grep -v "at::native\|rocclr\|elementwise" | tee $OUT/summary.txt
tail -3 $OUT/pmc3.log
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
