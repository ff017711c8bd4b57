# rocprofv3 trace + PMC passes with TEN frames per dispatch (r06_final's profile ran one frame per dispatch: an explicit --clip 10 did not divide the 64 resident sets)
bash tools/profile_pmc.sh r06_final 2>&1 | grep -v "at::native" | head -70
python3 tools/traffic_json.py gpurun_out/prof_r06_final $O/r06_c2_traffic.json 10
cp $O/r06_c2_traffic.json profiles/r06_c2_traffic.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 1500 $O/bench_driver.json
