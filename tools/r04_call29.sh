#!/bin/bash
# round 4, GPU call 29: the LUT samplers with their new defaults (waves per SIMD / tap rows in flight), ahead of time too; the GPU tier; the driver command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zc; mkdir -p $O
b() { name=$1; shift; timeout 400 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frac"), d["config"]["backend"], d["config"].get("parity_vs_oracle"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
b bicubic --interp 4
b lanczos --interp 8
b nv12_bicubic --fmt NV12 --interp 4
b nv12_lanczos --fmt NV12 --interp 8
b p010_lanczos --fmt P010LE --interp 8
b yuv420p_lanczos --fmt YUV420P --interp 8
b aot_bicubic --interp 4 --jit 0 --clip 1
b aot_lanczos --interp 8 --jit 0 --clip 1
b aot_nv12_lanczos --fmt NV12 --interp 8 --jit 0 --clip 1
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -3 $O/gpu_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; d=json.load(open('$O/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_frame'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config']['parity_vs_oracle'])" | tee -a $O/summary.txt
