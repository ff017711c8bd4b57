#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 900 python -m pytest tests/test_gpu_pass1.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -s > gpurun_out/r02g/pytest.log 2>&1
echo "pytest rc=$?" | tee gpurun_out/r02g/summary.txt
grep -i "certified second\|passed\|failed\|Error" gpurun_out/r02g/pytest.log | tail -8
for v in 0 4; do for res in 64; do
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --variant $v --resident $res > gpurun_out/r02g/bench_v$v.json 2> gpurun_out/r02g/bench_v$v.err
python3 -c "import json; d=json.load(open('gpurun_out/r02g/bench_v$v.json')); print('variant',$v,d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])"
done; done
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 TMPDIR=/tmp
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --resident 16"
OUT=gpurun_out/r02g/prof; mkdir -p $OUT
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python3 tools/summarize_prof.py $OUT 2>&1 | grep -v "at::native\|rocclr\|elementwise" | tee $OUT/summary.txt
rm -rf $OUT/pmc1 $OUT/pmc2
