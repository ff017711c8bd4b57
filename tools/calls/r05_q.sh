# where a wave's life goes inside the branch-free row: GFW_TIMELINE=2 (per-block shader clocks), C2 and C2 Lanczos4
GFW_TIMELINE_FILE=$O/tl_c2.bin GFW_JIT_DEFS="GFW_TIMELINE=2" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/bench_tl.json 2> $O/bench_tl.err
python3 tools/analyze_blocks.py $O/tl_c2.bin.blocks 2048
python3 tools/analyze_timeline.py $O/tl_c2.bin 2048 | head -8
GFW_TIMELINE_FILE=$O/tl_c2_l4.bin GFW_JIT_DEFS="GFW_TIMELINE=2" timeout 300 python3 bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-parity --interp 8 > $O/bench_tl8.json 2> $O/bench_tl8.err
python3 tools/analyze_blocks.py $O/tl_c2_l4.bin.blocks 1536
