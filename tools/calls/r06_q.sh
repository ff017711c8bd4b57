#!/bin/bash
# round 6, call q: what paces the per-plane EWA kernel — instruction counts and wait classes (1080p NV12, interpolation 10), PMC passes in their own runs
O=gpurun_out/r06_q; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 10 --warmup 2 --c1 --interp 10 --no-cpu-baseline --no-parity"
pmc() { n=$1; shift; timeout 200 rocprofv3 -f csv --pmc "$@" -d $O/pmc$n -o pmc$n -- $CMD > $O/bench_pmc$n.log 2>&1; }
pmc 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
pmc 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pmc 3 SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM
mkdir -p $O/x; cp -r $O/pmc1 $O/pmc2 $O/pmc3 $O/x/ 2>/dev/null
python3 tools/summarize_prof.py $O/x > $O/summary.txt 2>&1
cat $O/summary.txt | grep -v "^$" | head -80
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/x
