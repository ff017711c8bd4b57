# C5 with sixteen frames per launch (the finisher and its gaps amortised over twice the frames)
rec() { local name="$1"; shift; timeout 200 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
rec c5_clip16 --no-cpu-baseline --c5 --frames 3200 --clip 16
