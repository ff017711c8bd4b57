#!/bin/bash
# round 4, GPU call 13: C5 with the checksums on a second stream; the GPU tier on the library built without SLP ahead of time as well
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frac"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-900:])
PY
}
b c5 --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
b c5_again --gpus 1 --c5 --frames 2000 --warmup 16 --no-cpu-baseline
b aot_frame --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --clip 1 --jit 0
b driver --gpus 1 --steps 20 --warmup 5
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
