# the round's records: the whole GPU suite (serial, and under -n 4), the multi-device program beside other GPU work, every row of DESIGN.md section 4 from the FINAL
# library (each line bit-exact against the oracle in the same run), the driver's own command, rocprofv3 trace + PMC passes of the default workload (-> profiles/r06_*),
# the traffic file stamped with the kernel source id
date
timeout 1500 python3 -m pytest tests -q -m gpu -x --tb=long -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -4 $O/suite_serial.log | tee -a $O/summary.txt
timeout 900 python3 -m pytest tests -q -m gpu -n 4 -rf --tb=long -p no:cacheprovider > $O/suite_n4.log 2>&1; tail -3 $O/suite_n4.log | tee -a $O/summary.txt; grep -n "^FAILED\|^ERROR" $O/suite_n4.log | head
date
fails=0
for rep in $(seq 1 12); do
  pids=()
  for p in 0 1 2; do ( timeout 300 ./tests/cpp/test_multi_device 4 36 > $O/md_${rep}_$p.out 2> $O/md_${rep}_$p.err; echo $? > $O/md_${rep}_$p.rc ) & pids+=($!); done
  ( timeout 300 python3 -m pytest tests/test_gpu_pass1_sweep.py -q -m gpu -x -p no:cacheprovider > /dev/null 2>&1 ) & pids+=($!)
  wait "${pids[@]}"
  for p in 0 1 2; do rc=$(cat $O/md_${rep}_$p.rc); if [ "$rc" != "0" ]; then fails=$((fails+1)); echo "multi-device FAIL rep $rep proc $p rc $rc"; tail -6 $O/md_${rep}_$p.err; fi; rm -f $O/md_${rep}_$p.rc $O/md_${rep}_$p.out $O/md_${rep}_$p.err; done
done
echo "multi-device beside the audit sweep, final library: 12 repetitions x 3 instances: $fails failing" | tee -a $O/summary.txt
date
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -3 $O/bench_$name.err | grep -v amdgpu.ids; }
rec driver --steps 20 --warmup 5
rec c2_200 --no-cpu-baseline --steps 200
rec c2_clip1 --no-cpu-baseline --steps 200 --clip 1
rec c2_clip16 --no-cpu-baseline --steps 208 --clip 16
rec c2_aot --no-cpu-baseline --steps 200 --jit 0 --clip 1
rec c2_streams2 --no-cpu-baseline --steps 200 --streams 2
rec c2_bicubic --no-cpu-baseline --interp 4
rec c2_lanczos --no-cpu-baseline --interp 8 --steps 100
rec c2_superview --no-cpu-baseline --digital gopro_superview --steps 100
rec c2_poly5 --no-cpu-baseline --lens-model poly5 --steps 100
rec c2_gopro --no-cpu-baseline --lens-model gopro --steps 100
rec c2_gopro_lanczos --no-cpu-baseline --lens-model gopro --steps 100 --interp 8
rec c2_opencv_standard --no-cpu-baseline --lens-model opencv_standard --steps 100
rec nv12 --no-cpu-baseline --fmt NV12
rec nv12_bicubic --no-cpu-baseline --fmt NV12 --interp 4
rec nv12_lanczos --no-cpu-baseline --fmt NV12 --interp 8 --steps 100
rec yuv420p --no-cpu-baseline --fmt YUV420P
rec p010 --no-cpu-baseline --fmt P010LE
rec p010_bicubic --no-cpu-baseline --fmt P010LE --interp 4
rec p010_lanczos --no-cpu-baseline --fmt P010LE --interp 8 --steps 100
rec c1 --no-cpu-baseline --c1
rec c3 --no-cpu-baseline --width 7680 --height 4320 --resident 16 --steps 96
rec c4_rgbaf --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 --steps 96
rec c4_gbrapf32 --no-cpu-baseline --fmt GBRAPF32LE --crop --resident 16 --steps 96
rec rgbaf16 --no-cpu-baseline --fmt RGBAF16 --resident 16 --steps 96
rec c5_10000 --no-cpu-baseline --c5 --frames 10000
rec per_plane --no-cpu-baseline --per-plane --steps 200
rec per_plane_clip1 --no-cpu-baseline --per-plane --steps 200 --clip 1
rec per_plane_frame_sync --no-cpu-baseline --per-plane --steps 200 --clip 1 --frame-sync
rec host --no-cpu-baseline --host-buffers --steps 40
GFW_NO_HIPRTC=1 timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --steps 200 > $O/bench_c2_nohiprtc.json 2>/dev/null; python3 -c "import json; d=json.load(open('$O/bench_c2_nohiprtc.json')); print('c2_nohiprtc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit'])" | tee -a $O/summary.txt
GFW_FORCE_DIST=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err; tail -c 300 $O/bench_rccl_1rank.json
date
bash tools/profile_pmc.sh r06_final 2>&1 | grep -v "at::native" | head -70
python3 tools/traffic_json.py gpurun_out/prof_r06_final $O/r06_c2_traffic.json 10
rec driver2 --steps 20 --warmup 5
date
