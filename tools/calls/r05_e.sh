# lean flush + packed rows + no tiny guard (first line, against 42.9 us of call d), then the ablations of the specialised kernel (1 no first pass, 2 no luma taps, 4 no chroma, 8 no projection)
bench GFW_JIT_DEFS=GFW_UNUSED_TAG=3 --steps 200
bench A=1 --steps 200
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=1 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=2 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=4 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=6 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=8 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=9 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=15 --steps 200 --no-parity
bench A=1 --interp 4
bench A=1 --interp 8
bench A=1 --fmt NV12
timeout 600 python3 -m pytest tests/test_gpu_pass1.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
