#!/bin/bash
# round 3, twenty-fifth GPU call: first-pass staging in half the LDS (u16 rows, fixed queue): eight workgroups per CU for 4:2:0 — whole suite, then the formats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
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
b nv12 $A --fmt NV12
b yuv420p $A --fmt YUV420P
b p010 $A --fmt P010
b nv12_lanczos $S --fmt NV12 --interp 8
b default200 $A
b driver --gpus 1 --steps 20 --warmup 5
b c1 $A --c1
b lanczos $S --interp 8
b aot_nv12 $A --fmt NV12 --jit 0 --clip 1
