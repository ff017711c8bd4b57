# where the time is (ablations of the specialised kernel: 1 no first pass, 2 no luma taps, 4 no chroma, 8 no projection), and the tile height (lane-rows per tile) of the lattice form
bench GFW_JIT_DEFS=GFW_UNUSED_TAG=2 --steps 200
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=1 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=2 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=4 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=6 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=8 --steps 200 --no-parity
bench GFW_JIT_DEFS=GFW_ABLATE_FORCE=15 --steps 200 --no-parity
bench "GFW_JIT_RB_FAST=5 GFW_JIT_DEFS=GFW_UNUSED_TAG=2" --steps 200
bench "GFW_JIT_RB_FAST=6 GFW_JIT_DEFS=GFW_UNUSED_TAG=2" --steps 200
bench "GFW_JIT_RB_FAST=8 GFW_JIT_DEFS=GFW_UNUSED_TAG=2" --steps 200
bench "GFW_JIT_RB_FAST=9 GFW_JIT_DEFS=GFW_UNUSED_TAG=2" --steps 200
bench "GFW_JIT_RB_FAST=6 GFW_JIT_DEFS=GFW_UNUSED_TAG=2" --steps 200 --clip 16
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=2" --steps 200 --clip 16
bench "GFW_JIT_DEFS=GFW_P1_LATTICE_MAX_E=0.08f" --steps 200
