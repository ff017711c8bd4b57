#!/bin/bash
# next round, first GPU call: the specialised kernel with and without LLVM's SLP vectoriser (profiles/r03_slp_static.txt) — the option travels through GFW_JIT_DEFS
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
b() { name=$1; defs=$2; shift 2; GFW_JIT_DEFS="$defs" timeout 120 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline", {})
    print("$name [$defs]", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("parity_vs_reference_kernel", "")[:9])
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-400:])
PY
}
for rep in 1 2; do
  b c2_default_$rep "GFW_NOOP=1"
  b c2_noslp_$rep "-fno-slp-vectorize"
done
b lanczos_default "GFW_NOOP=1" --interp 8 --steps 64 --warmup 16
b lanczos_noslp "-fno-slp-vectorize" --interp 8 --steps 64 --warmup 16
b bicubic_default "GFW_NOOP=1" --interp 4
b bicubic_noslp "-fno-slp-vectorize" --interp 4
b nv12_default "GFW_NOOP=1" --fmt NV12
b nv12_noslp "-fno-slp-vectorize" --fmt NV12
b gopro_default "GFW_NOOP=1" --lens-model gopro
b gopro_noslp "-fno-slp-vectorize" --lens-model gopro
b superview_default "GFW_NOOP=1" --digital gopro_superview
b superview_noslp "-fno-slp-vectorize" --digital gopro_superview
GFW_JIT_DEFS="-fno-slp-vectorize" timeout 300 python -m pytest tests/test_gpu_jit.py -m gpu -q -x -p no:cacheprovider > $O/jit_noslp.log 2>&1; echo "jit tests under -fno-slp-vectorize rc $?" | tee -a $O/summary.txt; tail -3 $O/jit_noslp.log
