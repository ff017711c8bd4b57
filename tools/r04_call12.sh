#!/bin/bash
# round 4, GPU call 12: cheap experiments on the C2 kernel — rows per loop iteration, scheduler strategies, grid size
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
b() { name=$1; defs=$2; shift 2; GFW_JIT_DEFS="$defs" timeout 150 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs]", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"]["jit"]["state"])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-500:])
PY
}
b base "GFW_NOOP=1"
b rows2 "GFW_ROW_UNROLL=2"
b rows4 "GFW_ROW_UNROLL=4"
b maxilp "-mllvm;-amdgpu-sched-strategy=max-ilp"
b maxmem "-mllvm;-amdgpu-sched-strategy=max-memory-clause"
b o2 "-O2"
b grid4096 "GFW_NOOP=2" --grid 4096
b grid3072 "GFW_NOOP=3" --grid 3072
b grid1024 "GFW_NOOP=4" --grid 1024
b base2 "GFW_NOOP=5"
b nv12_rows2 "GFW_ROW_UNROLL=2" --fmt NV12
b nv12_base "GFW_NOOP=1" --fmt NV12
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_noslp.so b aot_noslp "GFW_NOOP=1" --jit 0 --clip 1
b aot "GFW_NOOP=1" --jit 0 --clip 1
