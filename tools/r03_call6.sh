#!/bin/bash
# round 3, sixth GPU call: parity of the second set of trims + clip fast path, benches of every configuration, PMC of the shipped library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -8 $O/gputests.log
timeout 300 python -m pytest tests/test_gpu_ref_opencl.py -k libgfwarp -m gpu -q -s -p no:cacheprovider 2>&1 | grep "identical\|passed\|failed" | cut -c1-300
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
b driver2 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b default200 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline
b jit_frame --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --clip 1
b aot_frame --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --clip 1 --jit 0
b lanczos --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
b bicubic --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 4
b c1 --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --c1
b c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
b c4 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --fmt RGBAF32 --crop --resident 16
b c4planar --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --fmt GBRAPF32LE --crop --resident 16
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
b superview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_superview
bash tools/profile_r03.sh r03b > $O/profile.log 2>&1; grep -v "at::native\|rocclr\|^W2026" $O/profile.log | tail -40
