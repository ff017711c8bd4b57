#!/bin/bash
# round 3, seventeenth GPU call: mesh through the specialised kernel; same-address atomics of the checksum kernel (block cap sweep on C5);
# clip launches dealt to two streams; waves for the blend bodies
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_jit.py tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider -k "mesh or checksum64 or generic_features" > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -5 $O/gputests.log
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
for cb in 2048 1024 512 256 128; do GFW_CHECKSUM_BLOCKS=$cb b c5_cb$cb --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline; done
b c2_s1 $A
b c2_s2 $A --streams 2
b c2_s2_20 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --streams 2
b c2_s1_20 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b lanczos_s2 $S --interp 8 --streams 2
GFW_JIT_WAVES=8 b gopro_lca_w8 $S --lens-model gopro --lca 0.5
GFW_JIT_WAVES=7 b gopro_lca_w7 $S --lens-model gopro --lca 0.5
GFW_JIT_WAVES=7 b fisheye_lca_w7 $S --lca 0.5
GFW_JIT_WAVES=8 b superview_lca_w8 $S --digital gopro_superview --lca 0.6
GFW_JIT_WAVES=7 b gopro_lanczos_w7 $S --lens-model gopro --interp 8
