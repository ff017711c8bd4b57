#!/bin/bash
# round 4, GPU call 10: 16-bit integer-dot taps A/B, clip launches of 16 frames, Lanczos4 with the conditional second fetch back for 16-bit planes, kernel cache tests, GPU tier
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
b() { name=$1; defs=$2; shift 2; GFW_JIT_DEFS="$defs" timeout 150 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs]", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"]["jit"]["compile_ms"], d["config"]["jit"]["state"])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-800:])
PY
}
b c2 ""
b c2_dot16 "GFW_DOT16=1"
b c2_again ""
b c2_dot16_again "GFW_DOT16=1"
b c2_clip16 "" --clip 16 --resident 64
b c2_dot16_clip16 "GFW_DOT16=1" --clip 16 --resident 64
GFW_NO_HIPRTC=1 b c2_no_hiprtc ""
b lanczos "" --interp 8 --steps 64 --warmup 16
b bicubic "" --interp 4
b p010 "" --fmt P010LE
b p010_dot16 "GFW_DOT16=1" --fmt P010LE
timeout 600 python -m pytest tests/test_gpu_jit_cache.py -m gpu -q -p no:cacheprovider > $O/cache_tests.log 2>&1; echo "cache tests rc $?" | tee -a $O/summary.txt; tail -8 $O/cache_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
