# aligned bilinear tap fetches (GFW_ALIGNED_TAPS): A/B in one process compiler, shipped kernels of the 16-bit formats; TA / TCP counters of the C2 kernel
bench "GFW_JIT_DEFS=GFW_ALIGNED_TAPS=1" --steps 200
bench "GFW_JIT_DEFS=GFW_ALIGNED_TAPS=0" --steps 200
bench "GFW_JIT_DEFS=GFW_ALIGNED_TAPS=1" --steps 200
bench "GFW_JIT_DEFS=GFW_ALIGNED_TAPS=0" --steps 200
bench A=1 --steps 200
bench A=1 --fmt P010LE
bench A=1 --fmt YUV444P16LE
bench A=1 --width 7680 --height 4320
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
rocprofv3 -L > $O/counters.txt 2>&1
CMD="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-parity"
pmc() { n=$1; shift; timeout 120 rocprofv3 -f csv --pmc "$@" -d $O/pmc$n -o pmc$n -- $CMD > $O/bench_pmc$n.log 2>&1; }
pmc 6 TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE
pmc 7 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pmc 8 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pmc 9 TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
python3 tools/summarize_prof.py $O 2>&1 | grep -v "at::native" | grep -A12 "gfw_jit_kernel" | head -70
rm -rf $O/pmc6 $O/pmc7 $O/pmc8 $O/pmc9
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py tests/test_gpu_pass1.py -x -q -m gpu 2>&1 | tail -4
