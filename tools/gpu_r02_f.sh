#!/bin/bash
cd $GRAFT_REPO_ROOT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 TMPDIR=/tmp
for v in ${VARIANTS:-159 47}; do
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --resident 8 --variant $v"
OUT=gpurun_out/r02f/v$v; mkdir -p $OUT
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 -f csv --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_DCACHE_MISSES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
echo "=== variant $v"; python3 tools/summarize_prof.py $OUT 2>&1 | grep -v "at::native\|rocclr\|elementwise" | tee $OUT/summary.txt
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
done
