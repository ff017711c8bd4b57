# ablations of the specialised kernel ON the branch-free lane-row (1 no first pass, 2 no luma taps, 4 no chroma, 8 no projection, 16 no luma store, 32 no per-lane matrix fetch)
for a in 0 1 2 4 6 8 16 18 22 32 33 40 63; do bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=$a --steps 200 --no-parity; done
bench "GFW_JIT_DEFS=-mllvm;-structurizecfg-skip-uniform-regions" --steps 200
bench "GFW_JIT_WAVES=7" --steps 200
bench "GFW_JIT_WAVES=6" --steps 200
