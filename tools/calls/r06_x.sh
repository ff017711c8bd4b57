#!/bin/bash
# round 6, call x: why are 8K launches of sixteen frames slower than launches of four?  HBM traffic and wave counters of both (PMC passes in their own runs)
O=gpurun_out/r06_x; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 64 --warmup 16 --clip 16 --width 7680 --height 4320 --resident 16 --no-cpu-baseline --no-parity"
for mode in cap4 cap16; do
  if [ $mode = cap16 ]; then export GFW_CLIP_LAUNCH_MB=8000; else unset GFW_CLIP_LAUNCH_MB; fi
  pmc() { n=$1; shift; timeout 200 rocprofv3 -f csv --pmc "$@" -d $O/$mode/pmc$n -o pmc$n -- $CMD > $O/bench_${mode}_pmc$n.log 2>&1; }
  pmc 1 FETCH_SIZE
  pmc 2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  pmc 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
  pmc 4 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  echo "== $mode" | tee -a $O/summary.txt
  python3 tools/summarize_prof.py $O/$mode 2>&1 | grep -v "^$" | grep -A14 "gfw_jit_kernel" | tee -a $O/summary.txt
  rm -rf $O/$mode
done
