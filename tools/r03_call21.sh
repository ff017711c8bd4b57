#!/bin/bash
# round 3, twenty-first GPU call: the tail of a launch handed out wave by wave from per-XCD counters (GFW_DYN_TAIL sixteenths)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_fullsize.py tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider -x > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -4 $O/gputests.log
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
A="--gpus 1 --steps 208 --warmup 16 --no-cpu-baseline"
S="--gpus 1 --steps 64 --warmup 16 --no-cpu-baseline"
for dt in 0 2 3 4 6 16; do GFW_JIT_DEFS="GFW_DYN_TAIL=$dt" b c2_dt$dt $A; done
GFW_DYN_TAIL=0 b c2_dtoff $A
b c2_s2 $A --streams 2
b jit_frame $A --clip 1
GFW_DYN_TAIL=0 b jit_frame_off $A --clip 1
b driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b lanczos $S --interp 8
GFW_DYN_TAIL=0 b lanczos_off $S --interp 8
b c1 $A --c1
GFW_DYN_TAIL=0 b c1_off $A --c1
b c4 $S --fmt RGBAF32 --crop --resident 16
GFW_DYN_TAIL=0 b c4_off $S --fmt RGBAF32 --crop --resident 16
GFW_JIT_DEFS="GFW_TIMELINE=1" GFW_TIMELINE_FILE=$O/tl_c2.bin timeout 300 python bench.py $A --no-parity > $O/bench_tl.json 2> $O/bench_tl.err
python tools/analyze_timeline.py $O/tl_c2.bin 2048 > $O/timeline_c2_dyn.txt; grep -E "busy|span|percentiles|per-SIMD|units" $O/timeline_c2_dyn.txt
