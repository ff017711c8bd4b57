#!/bin/bash
# round 4, first GPU call: (a) the specialised kernel with and without LLVM's SLP vectoriser (profiles/r03_slp_static.txt; option through GFW_JIT_DEFS);
# (b) the issue-class PMC passes VERDICT r03 asks for (VMEM issue cycles, LDS waits, per-class VALU counts), counters in their own runs, no trace domains with --pmc.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; defs=$2; shift 2; GFW_JIT_DEFS="$defs" timeout 120 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs]", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("parity_vs_reference_kernel", "")[:9])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-400:])
PY
}
for rep in 1 2; do
  b c2_default_$rep "GFW_NOOP=1"
  b c2_noslp_$rep "-fno-slp-vectorize"
done
b lanczos_default "GFW_NOOP=1" --interp 8 --steps 64 --warmup 16
b lanczos_noslp "-fno-slp-vectorize" --interp 8 --steps 64 --warmup 16
b bicubic_default "GFW_NOOP=1" --interp 4
b bicubic_noslp "-fno-slp-vectorize" --interp 4
b nv12_default "GFW_NOOP=1" --fmt NV12
b nv12_noslp "-fno-slp-vectorize" --fmt NV12
rocprofv3 -L > $O/counters_avail.txt 2>&1
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
P=$O/prof; mkdir -p $P
CMD="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-parity"
pmc() { n=$1; shift; timeout 60 rocprofv3 -f csv --pmc "$@" -d $P/pmc$n -o pmc$n -- $CMD > $P/bench_pmc$n.log 2>&1; echo "pmc$n rc $?"; }
pmc 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pmc 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY
pmc 3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64
pmc 4 SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT
pmc 5 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
python3 tools/summarize_prof.py $P > $P/summary_all.txt 2>&1
cat $P/summary_all.txt | grep -v "^$" | head -80
for n in 1 2 3 4 5; do tail -3 $P/bench_pmc$n.log | cut -c1-300; done
rm -rf $P/pmc1 $P/pmc2 $P/pmc3 $P/pmc4 $P/pmc5
