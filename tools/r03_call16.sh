#!/bin/bash
# round 3, sixteenth GPU call: run-time specialisation of the generic-model body (every lens model / feature bit) — whole suite, then
# specialised against ahead-of-time on other lens models; the unrolled checksum (C5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -8 $O/gputests.log
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
S="--gpus 1 --steps 64 --warmup 8 --no-cpu-baseline"
b driver --gpus 1 --steps 20 --warmup 5
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
for m in opencv_standard poly5 sony gopro insta360; do
  b ${m}_jit $S --lens-model $m
  b ${m}_aot $S --lens-model $m --jit 0 --clip 1
done

GFW_JIT_WAVES=7 b gopro_w7 $S --lens-model gopro
GFW_JIT_WAVES=6 b gopro_w6 $S --lens-model gopro
b fisheye_lca_jit $S --lca 0.5
b fisheye_lca_aot $S --lca 0.5 --jit 0 --clip 1
GFW_JIT_WAVES=8 b fisheye_lca_w8 $S --lca 0.5
b gopro_lca_jit $S --lens-model gopro --lca 0.5
b gopro_lca_aot $S --lens-model gopro --lca 0.5 --jit 0 --clip 1
b gopro_lanczos_jit $S --lens-model gopro --interp 8
b gopro_lanczos_aot $S --lens-model gopro --interp 8 --jit 0 --clip 1
b superview_lca_jit $S --digital gopro_superview --lca 0.6
b superview_lca_aot $S --digital gopro_superview --lca 0.6 --jit 0 --clip 1
