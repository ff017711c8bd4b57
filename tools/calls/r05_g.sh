# deferred stores (a lane-row's stores leave behind the next row's matrix fetches): A/B in one process compiler, the shipped kernel, what the ablations say now
bench "GFW_JIT_DEFS=GFW_STORE_DEFER=1" --steps 200
bench "GFW_JIT_DEFS=GFW_STORE_DEFER=0" --steps 200
bench "GFW_JIT_DEFS=GFW_STORE_DEFER=1" --steps 200
bench "GFW_JIT_DEFS=GFW_STORE_DEFER=0" --steps 200
bench A=1 --steps 200
bench A=1
for a in 1 4 16 32; do bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=$a --steps 200 --no-parity; done
bench "GFW_JIT_WAVES=7" --steps 200
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
