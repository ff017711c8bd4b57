#!/bin/bash
# tools/profile_r03.sh <tag> — rocprofv3 kernel-trace stats + PMC passes of the SHIPPED library under bench.py's default workload (C2, clip calls of
# 8 frames through the run-time specialised kernel: one dispatch of gfw_jit_kernel = 8 frames).  Counters in their own runs (no trace domains with --pmc).
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1   # bench.py as a worker itself (no launcher process between rocprofv3 and the kernels)
CMD="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-parity ${BENCH_EXTRA:-}"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
timeout 120 rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/bench_pmc1.log 2>&1
timeout 120 rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/bench_pmc2.log 2>&1
timeout 120 rocprofv3 -f csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/bench_pmc3.log 2>&1
timeout 120 rocprofv3 -f csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/bench_pmc4.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
tail -2 $OUT/bench_trace.log | cut -c1-400
mkdir -p $OUT/keep; find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/keep/ \;
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
