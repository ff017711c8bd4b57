#!/bin/bash
# round 3, seventh GPU call: the whole suite again after restoring the generic-model branch's validity tests; the generic-model benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputests.log 2>&1; echo "pytest rc $?"; tail -8 $O/gputests.log
timeout 300 python -m pytest tests/test_gpu_ref_opencl.py -k libgfwarp -m gpu -q -s -p no:cacheprovider 2>&1 | grep "identical\|passed\|failed" | cut -c1-300
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frames_per_launch"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("jit", {}).get("compile_ms"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
b superview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_superview
b driver --gpus 1 --steps 20 --warmup 5
