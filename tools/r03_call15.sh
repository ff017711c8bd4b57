#!/bin/bash
# round 3, fifteenth GPU call: dynamic unit distribution A/B, then rocprofv3 + PMC of the shipped defaults
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
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
b base $A
GFW_JIT_DEFS="GFW_DYN_UNITS=1;GFW_PRIO_MODE=0" b dyn_prio0 $A
GFW_JIT_DEFS="GFW_DYN_UNITS=1" b dyn_prio1 $A
GFW_JIT_DEFS="GFW_DYN_UNITS=1;GFW_PRIO_MODE=0" b dyn_prio0_clip1 $A --clip 1
b base_clip1 $A --clip 1
GFW_JIT_DEFS="GFW_DYN_UNITS=1;GFW_PRIO_MODE=0" b dyn_prio0_c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
GFW_JIT_DEFS="GFW_DYN_UNITS=1;GFW_PRIO_MODE=0" b dyn_prio0_l8 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
GFW_JIT_DEFS="GFW_DYN_UNITS=1;GFW_PRIO_MODE=0" b dyn_prio0_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b base2 $A
bash tools/profile_r03.sh r03c > $O/profile.log 2>&1; grep -v "at::native\|rocclr\|^W2026" $O/profile.log | tail -36
