#!/bin/bash
# round 4, GPU call 3: the branch-free lane-row (GFW_FASTROW) against the divergent per-pixel code, joint / sequential pixels, 8 / 7 / 6 waves per SIMD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
b() { name=$1; defs=$2; waves=$3; shift 3; GFW_JIT_WAVES=$waves GFW_JIT_DEFS="$defs" timeout 120 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs] waves=$waves", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("parity_vs_reference_kernel", "")[:9])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
b old "GFW_FASTROW=0" 0
b joint8 "GFW_NOOP=1" 0
b joint7 "GFW_NOOP=1" 7
b joint6 "GFW_NOOP=1" 6
b seq8 "GFW_FASTROW_JOINT=0" 0
b seq7 "GFW_FASTROW_JOINT=0" 7
b joint8_noslp "-fno-slp-vectorize" 0
b seq8_noslp "GFW_FASTROW_JOINT=0;-fno-slp-vectorize" 0
b old2 "GFW_FASTROW=0" 0
b nv12_old "GFW_FASTROW=0" 0 --fmt NV12
b nv12_joint "GFW_NOOP=1" 0 --fmt NV12
b nv12_seq "GFW_FASTROW_JOINT=0" 0 --fmt NV12
