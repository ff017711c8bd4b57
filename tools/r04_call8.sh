#!/bin/bash
# round 4, GPU call 8: the branch-free row for the LUT samplers, stretched clips through the fused kernel, the GPU tier
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
b() { name=$1; defs=$2; shift 2; GFW_JIT_DEFS="$defs" timeout 150 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs]", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("parity_vs_reference_kernel", "")[:9])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-800:])
PY
}
b c2 "GFW_NOOP=1"
b lanczos "GFW_NOOP=1" --interp 8 --steps 64 --warmup 16
b lanczos_oldrow "GFW_FASTROW_LUT=0" --interp 8 --steps 64 --warmup 16
b bicubic "GFW_NOOP=1" --interp 4
b bicubic_oldrow "GFW_FASTROW_LUT=0" --interp 4
b nv12_lanczos "GFW_NOOP=1" --interp 8 --fmt NV12 --steps 64 --warmup 16
b lanczos_w7 "GFW_NOOP=1" --interp 8 --steps 64 --warmup 16
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
