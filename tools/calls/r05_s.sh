# interleaved-chroma LUT rows fetched group-first (NV12 / P010 bicubic, Lanczos4): shipped kernels
bench A=1 --fmt NV12 --interp 4
bench A=1 --fmt NV12 --interp 8 --steps 100
bench A=1 --fmt P010LE --interp 4
bench A=1 --fmt P010LE --interp 8 --steps 100
bench A=1 --fmt RGBA --steps 100 --interp 4
bench A=1 --steps 200
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py tests/test_gpu_fused_coverage.py tests/test_ref_golden.py -x -q -m gpu 2>&1 | tail -4
