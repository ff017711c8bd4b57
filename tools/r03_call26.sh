#!/bin/bash
# round 3, twenty-sixth (last) GPU call: A/B of the queued instruction trims of the baked bilinear path (GFW_TRIMS bit mask, run-time builds
# via GFW_JIT_DEFS), parity of the timed region's last frames against the oracle per variant; ~4 GPU-minutes were left for it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
A="--gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
for t in 0 31 16 8 4 2 1 27; do
  GFW_JIT_DEFS="GFW_TRIMS=$t" timeout 70 python bench.py $A > $O/bench_trims$t.json 2> $O/bench_trims$t.err
  python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_trims$t.json")); r = d.get("roofline", {})
    print("trims=$t", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"))
except Exception as e:
    print("trims=$t FAILED", e, open("$O/bench_trims$t.err").read()[-400:])
PY
done
GFW_JIT_DEFS="GFW_TRIMS=31" timeout 200 python -m pytest tests/test_gpu_jit.py -m gpu -q -x -p no:cacheprovider > $O/jit_trims31.log 2>&1; echo "jit tests under trims=31 rc $?" | tee -a $O/summary.txt; tail -3 $O/jit_trims31.log
