# the final tree: driver command, the rows that changed since tools/calls/r05_final.sh (planar float, RGBAf16, LUT samplers of 4:2:0), trace + PMC + traffic file
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; }
rec driver --steps 20 --warmup 5
rec c2_200 --no-cpu-baseline --steps 200
rec c2_bicubic --no-cpu-baseline --interp 4
rec c2_lanczos --no-cpu-baseline --interp 8 --steps 100
rec nv12 --no-cpu-baseline --fmt NV12
rec nv12_bicubic --no-cpu-baseline --fmt NV12 --interp 4
rec nv12_lanczos --no-cpu-baseline --fmt NV12 --interp 8 --steps 100
rec p010 --no-cpu-baseline --fmt P010LE
rec p010_bicubic --no-cpu-baseline --fmt P010LE --interp 4
rec p010_lanczos --no-cpu-baseline --fmt P010LE --interp 8 --steps 100
rec c3 --no-cpu-baseline --width 7680 --height 4320 --resident 16 --steps 100
rec c4_rgbaf --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 --steps 100
rec c4_gbrapf32 --no-cpu-baseline --fmt GBRAPF32LE --crop --resident 16 --steps 100
rec rgbaf16 --no-cpu-baseline --fmt RGBAF16 --steps 100
rec c5_10000 --no-cpu-baseline --c5 --frames 10000
bash tools/profile_pmc.sh r05_final2 2>&1 | grep -v "at::native" | head -60
python3 tools/traffic_json.py gpurun_out/prof_r05_final2 $O/r05_c2_traffic.json 8
