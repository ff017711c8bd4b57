# compiler-option scan of the specialised C2 kernel (entries of GFW_JIT_DEFS that start with '-' are passed to hiprtc as they are); each against the same-process default
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=11" --steps 200
bench "GFW_JIT_DEFS=-mllvm;-amdgpu-sched-strategy=max-ilp" --steps 200
bench "GFW_JIT_DEFS=-mllvm;-amdgpu-sched-strategy=max-memory-clause" --steps 200
bench "GFW_JIT_DEFS=-mllvm;-amdgpu-enable-max-ilp-scheduling-strategy" --steps 200
bench "GFW_JIT_DEFS=-O2" --steps 200
bench "GFW_JIT_DEFS=-mllvm;-amdgpu-early-inline-all=true" --steps 200
bench "GFW_JIT_DEFS=-mllvm;-enable-post-misched=0" --steps 200
bench "GFW_JIT_DEFS=-mllvm;-amdgpu-waitcnt-forcezero=0;-mllvm;-amdgpu-use-aa-in-codegen=true" --steps 200
bench "GFW_JIT_DEFS=GFW_PRIO_MODE=0" --steps 200
bench "GFW_JIT_DEFS=GFW_UNUSED_TAG=11" --steps 200
