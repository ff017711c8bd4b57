# the final tree: whole GPU suite, the driver command, C5 at 10 000 frames both ways, then trace + PMC + the traffic file of this library
timeout 900 python3 -m pytest tests -q -m gpu -x -n 4 2>&1 | tail -5 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec c5_10000_in_kernel --no-cpu-baseline --c5 --frames 10000
rec c5_10000_pass --no-cpu-baseline --c5 --frames 10000 --sum-pass
rec c2_200 --no-cpu-baseline --steps 200
bash tools/profile_pmc.sh r05_final3 2>&1 | grep -v "at::native" | head -60
python3 tools/traffic_json.py gpurun_out/prof_r05_final3 $O/r05_c2_traffic.json 8
rec driver --steps 20 --warmup 5
