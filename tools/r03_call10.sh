#!/bin/bash
# round 3, tenth GPU call: whole suite (digital lens with infinity-safe maps, whole-tile elision), A/B of the baked kernel through GFW_JIT_DEFS
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -8 $O/gputests.log
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
b base --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
GFW_JIT_DEFS="GFW_P3_SPLIT=1" b split --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
GFW_JIT_DEFS="GFW_PRIO_MODE=0" b prio0 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
GFW_JIT_DEFS="GFW_P3_SPLIT=1;GFW_PRIO_MODE=0" b split_prio0 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
b base2 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
b driver --gpus 1 --steps 20 --warmup 5
b superview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_superview
b hyperview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_hyperview
b c1 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --c1
b c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
