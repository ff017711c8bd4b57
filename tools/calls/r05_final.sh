# the round's records: every row of DESIGN.md section 4 from the FINAL library (each line bit-exact against the oracle in the same run), the driver's own command, the
# rocprofv3 trace + PMC passes of the default workload (-> profiles/r05_*), the traffic file stamped with the kernel source id
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), d['config'].get('backend'), d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; }
rec driver --steps 20 --warmup 5
rec c2_200 --no-cpu-baseline --steps 200
rec c2_nohiprtc_env --no-cpu-baseline --steps 200
rec c2_clip1 --no-cpu-baseline --steps 200 --clip 1
rec c2_aot --no-cpu-baseline --steps 200 --jit 0 --clip 1
rec c2_streams2 --no-cpu-baseline --steps 200 --streams 2
rec c2_bicubic --no-cpu-baseline --interp 4
rec c2_lanczos --no-cpu-baseline --interp 8 --steps 100
rec c2_superview --no-cpu-baseline --digital gopro_superview --steps 100
rec c2_poly5 --no-cpu-baseline --lens-model poly5 --steps 100
rec c2_gopro --no-cpu-baseline --lens-model gopro --steps 100
rec nv12 --no-cpu-baseline --fmt NV12
rec nv12_bicubic --no-cpu-baseline --fmt NV12 --interp 4
rec nv12_lanczos --no-cpu-baseline --fmt NV12 --interp 8 --steps 100
rec yuv420p --no-cpu-baseline --fmt YUV420P
rec p010 --no-cpu-baseline --fmt P010LE
rec p010_bicubic --no-cpu-baseline --fmt P010LE --interp 4
rec c1 --no-cpu-baseline --c1
rec c3 --no-cpu-baseline --width 7680 --height 4320 --resident 16 --steps 100
rec c4_rgbaf --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 --steps 100
rec c4_gbrapf32 --no-cpu-baseline --fmt GBRAPF32LE --crop --resident 16 --steps 100
rec c5_10000 --no-cpu-baseline --c5 --frames 10000
rec per_plane --no-cpu-baseline --per-plane --steps 200
rec per_plane_clip1 --no-cpu-baseline --per-plane --steps 200 --clip 1
rec per_plane_frame_sync --no-cpu-baseline --per-plane --steps 200 --clip 1 --frame-sync
GFW_NO_HIPRTC=1 timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --steps 200 > $O/bench_c2_nohiprtc.json 2>/dev/null; python3 -c "import json; d=json.load(open('$O/bench_c2_nohiprtc.json')); print('c2_nohiprtc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit'])" | tee -a $O/summary.txt
GFW_FORCE_DIST=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err; tail -c 300 $O/bench_rccl_1rank.json
bash tools/profile_pmc.sh r05_final 2>&1 | grep -v "at::native" | head -70
python3 tools/traffic_json.py gpurun_out/prof_r05_final $O/r05_c2_traffic.json 8
