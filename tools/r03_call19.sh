#!/bin/bash
# round 3, nineteenth GPU call: sub-band interleave across the XCDs (GFW_SUB_BANDS), clip launches of 16 frames, ordered overlap; whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -6 $O/gputests.log
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
A="--gpus 1 --steps 208 --warmup 16 --no-cpu-baseline"
S="--gpus 1 --steps 64 --warmup 16 --no-cpu-baseline"
for sb in 1 2 4 8; do GFW_JIT_DEFS="GFW_SUB_BANDS=$sb" b c2_sb$sb $A; done
b c2_clip16 $A --clip 16
GFW_JIT_DEFS="GFW_SUB_BANDS=1" b c2_clip16_sb1 $A --clip 16
b c2_clip16_s2 $A --clip 16 --streams 2
b c2_clip8_s2 $A --streams 2
b jit_frame $A --clip 1
GFW_JIT_DEFS="GFW_SUB_BANDS=1" b jit_frame_sb1 $A --clip 1
b aot_frame $A --clip 1 --jit 0
b driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b driver16 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --clip 16
b lanczos $S --interp 8
b lanczos16 $S --interp 8 --clip 16
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
b c5_16 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline --clip 16
