# micro-steps (one-rounding bins, idle min, t2 == 0 coordinates): in-process twice + shipped twice; NV12 / P010 / C3; parity suites
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=9" --steps 200
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=9" --steps 200
bench A=1 --steps 200
bench A=1 --steps 200
bench A=1 --fmt NV12
bench A=1 --fmt P010LE
bench A=1 --interp 4
bench A=1 --interp 8 --steps 100
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py tests/test_gpu_fused_coverage.py tests/test_ref_golden.py -x -q -m gpu 2>&1 | tail -4
