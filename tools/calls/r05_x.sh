# after moving the launches outside the group lock: the coalescing tests (incl. the four-thread one), the per-plane bench rows, then the whole GPU suite
timeout 600 python3 -m pytest tests/test_gpu_coalesce.py -q -x 2>&1 | tail -5 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; }
rec per_plane --no-cpu-baseline --per-plane --clip 1
rec per_plane_8 --no-cpu-baseline --per-plane --clip 8
rec per_plane_sync --no-cpu-baseline --per-plane --frame-sync
rec driver --steps 20 --warmup 5
timeout 900 python3 -m pytest tests -q -m gpu -x -n 4 2>&1 | tail -5 | tee -a $O/summary.txt
