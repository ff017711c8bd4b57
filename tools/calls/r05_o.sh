# the branch-free row for packed RGB(A) planes and planar float frames (C4): shipped cache (new) against GFW_FASTROW=0 builds in process
bench A=1 --fmt RGBAF32 --crop --resident 16 --steps 100
bench "GFW_JIT_DEFS=GFW_FASTROW=0" --fmt RGBAF32 --crop --resident 16 --steps 100
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=8" --fmt RGBAF32 --crop --resident 16 --steps 100
bench A=1 --fmt GBRAPF32LE --crop --resident 16 --steps 100
bench "GFW_JIT_DEFS=GFW_FASTROW=0" --fmt GBRAPF32LE --crop --resident 16 --steps 100
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=8" --fmt GBRAPF32LE --crop --resident 16 --steps 100
bench A=1 --fmt RGBA --steps 100
bench "GFW_JIT_DEFS=GFW_FASTROW=0" --fmt RGBA --steps 100
bench A=1 --fmt RGBA64 --steps 100
bench A=1 --fmt YUV444P16LE --steps 100
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py tests/test_gpu_fused_coverage.py tests/test_ref_golden.py -x -q -m gpu 2>&1 | tail -4
