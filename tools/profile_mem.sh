#!/bin/bash
# tools/profile_mem.sh <tag> — memory-pipeline PMC passes (TA / TCP / SQ issue) for the bench workload; counters only (no trace domains).
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
CMD="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity ${BENCH_EXTRA:-}"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 -f csv --pmc $line -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/bench_pmc$i.log 2>&1
done <<'EOC'
TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
EOC
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -l -i "error\|invalid" $OUT/bench_pmc*.log | head
cat $OUT/summary.txt
for d in $OUT/pmc*; do rm -rf $d; done
