#!/bin/bash
# round 6, after the kernel-source change of the pointer-test fix (new kernel source id): rocprofv3 trace + PMC passes of the shipped library, ten frames per dispatch;
# the traffic file stamped with the new id; the driver's bench line
O=gpurun_out/r06_final3; mkdir -p $O
bash tools/profile_pmc.sh r06_final3 2>&1 | grep -v "at::native" | head -70
python3 tools/traffic_json.py gpurun_out/prof_r06_final3 $O/r06_c2_traffic.json 10
cp $O/r06_c2_traffic.json profiles/r06_c2_traffic.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 1500 $O/bench_driver.json
timeout 300 python3 bench.py --gpus 1 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
