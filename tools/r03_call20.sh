#!/bin/bash
# round 3, twentieth GPU call: per-wave timeline of a clip launch of the specialised kernel (where does a second stream find its 5 %?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
A="--gpus 1 --steps 208 --warmup 16 --no-cpu-baseline --no-parity"
GFW_JIT_DEFS="GFW_TIMELINE=1" GFW_TIMELINE_FILE=$O/tl_c2.bin timeout 300 python bench.py $A > $O/bench_tl.json 2> $O/bench_tl.err; tail -c 300 $O/bench_tl.err
python tools/analyze_timeline.py $O/tl_c2.bin 2048 | tee $O/timeline_c2.txt
GFW_JIT_DEFS="GFW_TIMELINE=1" GFW_TIMELINE_FILE=$O/tl_c2_clip16.bin timeout 300 python bench.py $A --clip 16 > $O/bench_tl16.json 2> $O/bench_tl16.err
python tools/analyze_timeline.py $O/tl_c2_clip16.bin 2048 | tee $O/timeline_c2_clip16.txt
