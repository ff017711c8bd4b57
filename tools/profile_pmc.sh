#!/bin/bash
# tools/profile_pmc.sh <tag> — rocprofv3 kernel-trace stats + PMC passes of the SHIPPED library under bench.py's default workload (C2, clip calls of 10 frames through the
# run-time specialised kernel: one dispatch of gfw_jit_kernel = 10 frames).  Counters in their own runs (no trace domains with --pmc).
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1   # bench.py as a worker itself (no launcher process between rocprofv3 and the kernels)
CMD="python bench.py --steps 80 --warmup 10 --clip 10 --no-cpu-baseline --no-parity ${BENCH_EXTRA:-}"      # every dispatch of gfw_jit_kernel carries 10 frames (warm-up, pre-heat and timed region alike): the driver line's launch
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
pmc() { n=$1; shift; timeout 120 rocprofv3 -f csv --pmc "$@" -d $OUT/pmc$n -o pmc$n -- $CMD > $OUT/bench_pmc$n.log 2>&1; }
pmc 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
pmc 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pmc 3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_IFETCH SQ_LDS_BANK_CONFLICT
pmc 4 FETCH_SIZE
pmc 5 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
tail -2 $OUT/bench_trace.log | cut -c1-400
mkdir -p $OUT/keep; find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/keep/ \;
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5
