#!/bin/bash
# round 3, twenty-third GPU call: the round's final library — whole suite, every configuration's bench record
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -6 $O/gputests.log
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frac"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
A="--gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
S="--gpus 1 --steps 64 --warmup 16 --no-cpu-baseline"
b driver --gpus 1 --steps 20 --warmup 5
b default200 $A
b streams2 $A --streams 2
b jit_frame $A --clip 1
b aot_frame $A --clip 1 --jit 0
b lanczos $S --interp 8
b bicubic $S --interp 4
b c1 $A --c1
b c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
b c4 $S --fmt RGBAF32 --crop --resident 16
b c4planar $S --fmt GBRAPF32LE --crop --resident 16
b nv12 $A --fmt NV12
b superview $S --digital gopro_superview
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
b lens_poly5_jit $S --lens-model poly5
b lens_gopro_jit $S --lens-model gopro
b fisheye_lca_jit $S --lca 0.5
b gopro_lca_jit $S --lens-model gopro --lca 0.5
