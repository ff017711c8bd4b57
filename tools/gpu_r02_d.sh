#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_gpu_pass1.py tests/test_gpu_fullsize.py -m gpu -x -q -s > gpurun_out/r02d/pytest.log 2>&1
echo "pytest rc=$?" | tee gpurun_out/r02d/summary.txt
grep -i "certified second\|passed\|failed" gpurun_out/r02d/pytest.log | tail -5
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
python3 -c "import json; d=json.load(open('gpurun_out/r02d/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])"
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 TMPDIR=/tmp
CMD="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --resident 16"
OUT=gpurun_out/r02d/prof; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python3 tools/summarize_prof.py $OUT 2>&1 | grep -v "at::native\|rocclr\|elementwise" | tee $OUT/summary.txt
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2
