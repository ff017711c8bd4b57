#!/bin/bash
# round 4, GPU call 22: the round's final library — rocprofv3 trace + PMC passes of the driver workload, every configuration's bench record, the GPU tier
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
bash tools/profile_r04.sh r04final > $O/profile.log 2>&1; tail -45 $O/profile.log
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frac"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("host_enqueue_ms_per_step"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
A="--gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
S="--gpus 1 --steps 64 --warmup 16 --no-cpu-baseline"
b driver --gpus 1 --steps 20 --warmup 5
b default200 $A
b streams2 $A --streams 2
b per_plane $A --per-plane --clip 1
b per_plane_clip8 $A --per-plane --clip 8
b jit_frame $A --clip 1
b aot_frame $A --clip 1 --jit 0
GFW_NO_HIPRTC=1 b no_hiprtc $A
b lanczos $S --interp 8
b bicubic $S --interp 4
b c1 $A --c1
b c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
b c4 $S --fmt RGBAF32 --crop --resident 16
b c4planar $S --fmt GBRAPF32LE --crop --resident 16
b nv12 $A --fmt NV12
b nv12_lanczos $S --fmt NV12 --interp 8
b p010 $A --fmt P010LE
b yuv420p $A --fmt YUV420P
b superview $S --digital gopro_superview
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
b lens_poly5_jit $S --lens-model poly5
b lens_gopro_jit $S --lens-model gopro
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -3 $O/gpu_tests.log
