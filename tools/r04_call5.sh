#!/bin/bash
# round 4, GPU call 5: plane coalescing (tests + bench --per-plane), the lane-row after the vote / ok-mask trims, ahead-of-time kernels with and without SLP
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_coalesce.py -m gpu -q -x -p no:cacheprovider > $O/coalesce_tests.log 2>&1; echo "coalesce tests rc $?" | tee -a $O/summary.txt; tail -15 $O/coalesce_tests.log
b() { name=$1; shift; timeout 150 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("parity_vs_reference_kernel", "")[:9], "enq", d["config"]["host_enqueue_ms_per_step"])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-800:])
PY
}
b c2
b c2_again
b per_plane_1 --per-plane --clip 1
b per_plane_8 --per-plane --clip 8
b per_plane_1_hostmat --per-plane --clip 1 --upload-matrices
GFW_COALESCE_PLANES=0 b per_plane_off --per-plane --clip 1 --steps 40
GFW_JIT=0 b per_plane_1_aot --per-plane --clip 1 --jit 0
b frame_1 --clip 1
b aot_frame --jit 0 --clip 1
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_noslp.so b aot_frame_noslp --jit 0 --clip 1
b nv12_per_plane_8 --per-plane --clip 8 --fmt NV12
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
