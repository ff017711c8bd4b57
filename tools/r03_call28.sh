#!/bin/bash
# round 3, twenty-eighth (last) GPU call: bench.py's new reference-fixture check under the driver's arguments and ahead of time; then rocprofv3 trace +
# PMC passes of the FINAL library (the C2 kernel changed once more after profiles/r03_yuv_fused_summary.txt was taken: 16-bit first-pass rows)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 70 python -m pytest tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider -k "driver_command_line or ahead_of_time" > $O/bench_tests.log 2>&1; echo "pytest rc $?"; tail -3 $O/bench_tests.log
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
P=$O/prof; mkdir -p $P
CMD="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-parity"
timeout 40 rocprofv3 --kernel-trace --stats -f csv -d $P/trace -o trace -- $CMD > $P/bench_trace.log 2>&1
python3 tools/summarize_prof.py $P > $P/summary.txt 2>&1; mkdir -p $P/keep; find $P/trace -name "*kernel_stats.csv" -exec cp {} $P/keep/ \; ; rm -rf $P/trace
timeout 30 rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/pmc1 -o pmc1 -- $CMD > $P/bench_pmc1.log 2>&1
timeout 30 rocprofv3 -f csv --pmc FETCH_SIZE -d $P/pmc3 -o pmc3 -- $CMD > $P/bench_pmc3.log 2>&1
timeout 30 rocprofv3 -f csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $P/pmc4 -o pmc4 -- $CMD > $P/bench_pmc4.log 2>&1
python3 tools/summarize_prof.py $P >> $P/summary.txt 2>&1
timeout 30 rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS -d $P/pmc2 -o pmc2 -- $CMD > $P/bench_pmc2.log 2>&1
python3 tools/summarize_prof.py $P > $P/summary_all.txt 2>&1
cat $P/summary_all.txt | grep -v "^$" | head -60
rm -rf $P/pmc1 $P/pmc2 $P/pmc3 $P/pmc4
