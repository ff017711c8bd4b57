# micro-steps: one clamp in the lattice row, translation2d == 0 coordinates (in-process compiler twice, shipped cache twice)
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=7" --steps 200
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=7" --steps 200
bench A=1 --steps 200
bench A=1 --steps 200
