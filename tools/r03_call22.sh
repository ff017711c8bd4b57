#!/bin/bash
# round 3, twenty-second GPU call: tail scheduler with one counter per (XCD, strip) on lines of their own; chunked fetches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_jit.py -m gpu -q -p no:cacheprovider -x -k "clip or formats or geometry" > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -3 $O/gputests.log
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
GFW_DYN_TAIL=0 b c2_off $A
for dt in 1 2 3; do GFW_JIT_DEFS="GFW_DYN_TAIL=$dt" b c2_dt$dt $A; done
for dt in 1 2 3; do GFW_JIT_DEFS="GFW_DYN_TAIL=$dt;GFW_DYN_CHUNK=2" b c2_dt${dt}_c2 $A; done
GFW_JIT_DEFS="GFW_DYN_TAIL=2;GFW_DYN_CHUNK=4" b c2_dt2_c4 $A
b jit_frame $A --clip 1
GFW_JIT_DEFS="GFW_TIMELINE=1;GFW_DYN_TAIL=2" GFW_TIMELINE_FILE=$O/tl_c2.bin timeout 300 python bench.py $A --no-parity > $O/bench_tl.json 2> $O/bench_tl.err
python tools/analyze_timeline.py $O/tl_c2.bin 2048 > $O/timeline_c2_dyn.txt; grep -E "busy|span|percentiles|per-SIMD|units|by wave" $O/timeline_c2_dyn.txt
