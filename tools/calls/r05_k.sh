# the north-star's wavefront tap share, upper bound (ablation 64: the pair's second pixel's taps by DPP from the neighbouring lane); the reference's call contracts
bench "GFW_JIT_DEFS=GFW_ABLATE_FORCE=64" --steps 200 --no-parity
bench "GFW_JIT_DEFS=GFW_ABLATE_FORCE=0" --steps 200 --no-parity
bench "GFW_JIT_DEFS=GFW_ABLATE_FORCE=64" --steps 200 --no-parity
bench "GFW_JIT_DEFS=GFW_ABLATE_FORCE=0" --steps 200 --no-parity
bench A=1 --per-plane --steps 200
bench A=1 --per-plane --steps 200 --clip 1
bench A=1 --per-plane --steps 200 --clip 1 --frame-sync
bench GFW_COALESCE_PLANES=0 --per-plane --steps 200 --clip 1
bench A=1 --steps 200 --clip 1
bench A=1 --steps 200 --clip 16
bench A=1 --steps 200 --jit 0 --clip 1
