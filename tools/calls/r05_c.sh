# lattice first pass: where does the time go?  per-wave phase clocks of both forms, PMC of the default, then the GPU suite
tl() { GFW_TIMELINE_FILE=$O/tl_$1.bin GFW_JIT_DEFS="GFW_TIMELINE=1$2" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/bench_tl.json 2> $O/bench_tl.err
  echo "== timeline $1"; python3 tools/analyze_timeline.py $O/tl_$1.bin 2048 | head -8; }
tl lattice ""
tl perpixel ";GFW_P1_LATTICE=0"
bash tools/profile_pmc.sh r05c_lattice 2>&1 | grep -v "at::native" | head -60
timeout 1200 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8
