# colour-range fix fused; C5 with the checksums on a second stream; bicubic at six waves from the shipped (ROCm 7.2) cache; 8-rank launcher; full GPU suite
bench A=1 --steps 200
bench A=1 --interp 4
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=4" --interp 4
bench A=1 --c5 --frames 2000 --sum-stream 1
bench A=1 --c5 --frames 2000 --sum-stream 0
bench A=1 --c5 --frames 2000 --sum-stream 1 --grid 2048
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
