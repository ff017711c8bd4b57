#!/bin/bash
# round 3, eighth GPU call: Lanczos4 over f32 copies of the planes (parity, timing), the 1080p reference residual with the relative tolerance
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
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
b lanczos --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
b lanczos_frame --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8 --clip 1
b lanczos_aot --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8 --jit 0 --clip 1
b lanczos_nv12 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8 --fmt NV12
b lanczos_8k --gpus 1 --steps 32 --warmup 8 --no-cpu-baseline --interp 8 --width 7680 --height 4320 --resident 16
b driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o trace -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-parity --interp 8 > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec grep "gfw_" {} \; | cut -c1-200
rm -rf $O/trace
