# matrix rows through LDS (GFW_MATRIX_WINDOW): A/B in one process compiler; shipped kernels; counters of the memory path
bench "GFW_JIT_DEFS=GFW_MATRIX_WINDOW=1" --steps 200
bench "GFW_JIT_DEFS=GFW_MATRIX_WINDOW=0" --steps 200
bench "GFW_JIT_DEFS=GFW_MATRIX_WINDOW=1" --steps 200
bench "GFW_JIT_DEFS=GFW_MATRIX_WINDOW=0" --steps 200
bench A=1 --steps 200
bench A=1
bench A=1 --fmt NV12
bench A=1 --fmt P010LE
bench A=1 --width 7680 --height 4320
bench A=1 --interp 4
bench A=1 --interp 8
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
CMD="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-parity"
pmc() { n=$1; shift; timeout 120 rocprofv3 -f csv --pmc "$@" -d $O/pmc$n -o pmc$n -- $CMD > $O/bench_pmc$n.log 2>&1; }
pmc 7 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pmc 9 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
pmc 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
python3 tools/summarize_prof.py $O 2>&1 | grep -v "at::native" | grep -A12 "gfw_jit_kernel" | head -60
rm -rf $O/pmc7 $O/pmc9 $O/pmc1
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py tests/test_gpu_pass1.py -x -q -m gpu 2>&1 | tail -4
