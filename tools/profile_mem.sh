#!/bin/bash
# tools/profile_mem.sh <tag> — memory-pipeline PMC passes (TA / TCP / SQ issue) for the bench workload; counters only (no trace domains).
# At most TWO counters of a TA / TCP block per pass: a pass that asks for more dies with "Request exceeds the capabilities of the
# hardware to collect" and rocprofv3 then sits until it is killed (round 2 lost 15 GPU-minutes to three such passes) — hence the
# short timeout on every pass, and the grep that stops the script at the first refusal.
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
  timeout 75 rocprofv3 -f csv --pmc $line -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/bench_pmc$i.log 2>&1
  if grep -q "exceeds the capabilities" $OUT/bench_pmc$i.log; then echo "pass $i refused: $line"; break; fi
done <<'EOC'
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
EOC
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -l -i "error\|invalid" $OUT/bench_pmc*.log | head
cat $OUT/summary.txt
for d in $OUT/pmc*; do rm -rf $d; done
