#!/bin/bash
# round 4, GPU call 4: ballot-ranked first-pass queue + priority per tile, on top of the branch-free lane-row (all with -fno-slp-vectorize unless noted)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
b() { name=$1; defs=$2; waves=$3; shift 3; GFW_JIT_WAVES=$waves GFW_JIT_DEFS="$defs" timeout 120 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs] waves=$waves", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("parity_vs_reference_kernel", "")[:9])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
b new "-fno-slp-vectorize" 0
b new_priorows "-fno-slp-vectorize;GFW_PRIO_ROWS=1" 0
b new_seq "-fno-slp-vectorize;GFW_FASTROW_JOINT=0" 0
b new_w7 "-fno-slp-vectorize" 7
b new_slp "GFW_NOOP=1" 0
b new2 "-fno-slp-vectorize" 0
b nv12 "-fno-slp-vectorize" 0 --fmt NV12
b nv12_seq "-fno-slp-vectorize;GFW_FASTROW_JOINT=0" 0 --fmt NV12
b p010 "-fno-slp-vectorize" 0 --fmt P010LE
b yuv420p "-fno-slp-vectorize" 0 --fmt YUV420P
b c1 "-fno-slp-vectorize" 0 --c1
b c3 "-fno-slp-vectorize" 0 --width 7680 --height 4320 --steps 64
b bicubic "-fno-slp-vectorize" 0 --interp 4
b lanczos "-fno-slp-vectorize" 0 --interp 8 --steps 64 --warmup 16
timeout 600 python -m pytest tests/test_gpu_jit.py tests/test_gpu_pass1.py tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/summary.txt; tail -3 $O/tests.log
