# lattice first pass, first timing: C2 default / per-pixel form (GFW_P1_LATTICE=0) A/B in one process compiler, then parity suites
bench GFW_JIT_DEFS=GFW_UNUSED_TAG=1 --steps 200
bench GFW_JIT_DEFS=GFW_P1_LATTICE=0 --steps 200
bench GFW_JIT_DEFS=GFW_UNUSED_TAG=1 --steps 200
bench GFW_JIT_DEFS=GFW_P1_LATTICE=0 --steps 200
bench A=1 --steps 200
bench A=1 --fmt NV12
bench A=1 --width 7680 --height 4320
timeout 900 python3 -m pytest tests/test_gpu_pass1.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py -x -q -m gpu 2>&1 | tail -5
