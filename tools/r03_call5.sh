#!/bin/bash
# round 3, fifth GPU call: new tests (explained-residual reference check, certificate sweep, builder ring), clip launches of 32, baked LUT samplers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=8 > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -22 $O/gputests.log
timeout 300 python -m pytest tests/test_gpu_ref_opencl.py tests/test_gpu_pass1_sweep.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | grep "identical\|certified\|passed\|failed" | cut -c1-300
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frames_per_launch"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("jit", {}).get("compile_ms"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
b driver --gpus 1 --steps 20 --warmup 5
b default200 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
b clip8 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --clip 8
GFW_JIT_LUT=1 b lanczos_jit6 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
GFW_JIT_LUT=1 GFW_JIT_WAVES=7 b lanczos_jit7 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
GFW_JIT_LUT=1 b bicubic_jit6 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 4
GFW_JIT_LUT=1 GFW_JIT_WAVES=7 b bicubic_jit7 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 4
b bicubic_aot --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 4
GFW_JIT_WAVES=8 b default_w8 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
GFW_JIT_WAVES=6 b default_w6 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
