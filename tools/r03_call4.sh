#!/bin/bash
# round 3, fourth GPU call: parity of the trimmed kernels (32-bit lane offsets, first-pass trims, saturation, chroma coordinate from luma), benches, profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -6 $O/gputests.log
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
b jit_frame --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --clip 1
b aot_frame --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --clip 1 --jit 0
b c1 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --c1
b c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
b c4 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --fmt RGBAF32 --crop --resident 16
GFW_JIT_LUT=1 b lanczos_jit --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
b lanczos --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
bash tools/profile_r03.sh r03 > $O/profile.log 2>&1; tail -60 $O/profile.log
