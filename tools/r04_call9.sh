#!/bin/bash
# round 4, GPU call 9: the rewritten bicubic / Lanczos4 tap row (unconditional aligned fetch, 32-bit offsets, no zero-adds) at 1 / 2 / 4 rows in flight; kernel cache tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
b() { name=$1; defs=$2; shift 2; GFW_JIT_CACHE= GFW_JIT_DEFS="$defs" timeout 150 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs]", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"]["jit"])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-800:])
PY
}
b lanczos_u2 "" --interp 8 --steps 64 --warmup 16
b lanczos_u1 "GFW_TAP_ROW_UNROLL(I)=1" --interp 8 --steps 64 --warmup 16
b lanczos_u4 "GFW_TAP_ROW_UNROLL(I)=4" --interp 8 --steps 64 --warmup 16
b bicubic_u2 "" --interp 4
b bicubic_u4 "GFW_TAP_ROW_UNROLL(I)=4" --interp 4
b bicubic_u1 "GFW_TAP_ROW_UNROLL(I)=1" --interp 4
b nv12_lanczos "" --interp 8 --fmt NV12 --steps 64 --warmup 16
b c2 ""
GFW_NO_HIPRTC=1 b c2_no_hiprtc ""
timeout 600 python -m pytest tests/test_gpu_jit_cache.py -m gpu -q -p no:cacheprovider > $O/cache_tests.log 2>&1; echo "cache tests rc $?" | tee -a $O/summary.txt; tail -15 $O/cache_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
