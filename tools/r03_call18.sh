#!/bin/bash
# round 3, eighteenth GPU call: XCD bands rotated across the frames of a clip launch (GFW_BAND_ROT) against two streams
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_jit.py -m gpu -q -p no:cacheprovider -k "clip or formats" > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -3 $O/gputests.log
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
A="--gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
S="--gpus 1 --steps 64 --warmup 8 --no-cpu-baseline"
for r in 0 1 3 5; do GFW_JIT_DEFS="GFW_BAND_ROT=$r" b c2_rot$r $A; done
b c2_rot3_s2 $A --streams 2
GFW_JIT_DEFS="GFW_BAND_ROT=0" b c2_rot0_s2 $A --streams 2
b driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b lanczos $S --interp 8
b bicubic $S --interp 4
b c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
b c4 $S --fmt RGBAF32 --crop --resident 16
b nv12 $A --fmt NV12
b c1 $A --c1
GFW_JIT_DEFS="GFW_BAND_ROT=0" b c1_rot0 $A --c1
