#!/bin/bash
# round 3, ninth GPU call: digital-lens clips through the specialised kernel (parity + timing), the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
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
b superview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_superview
b superview_aot --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_superview --jit 0 --clip 1
b hyperview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_hyperview
b lanczos --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
b driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
