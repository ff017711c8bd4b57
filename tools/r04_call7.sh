#!/bin/bash
# round 4, GPU call 7: interop import, coalescing, RCCL smoke, the whole GPU tier; how the CPU baseline scales on this host
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_interop.py tests/test_gpu_coalesce.py tests/test_gpu_rccl_smoke.py -m gpu -q -p no:cacheprovider > $O/new_tests.log 2>&1; echo "new tests rc $?" | tee -a $O/summary.txt; tail -25 $O/new_tests.log
(nproc; lscpu | head -25; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; numactl -H 2>/dev/null | head -12) > $O/host.txt 2>&1
OMP_PROC_BIND=close OMP_PLACES=threads timeout 600 python tools/cpu_baseline_scan.py 3840 2160 8,16,32,64,128,256 > $O/cpu_scan_pinned.txt 2>&1; tail -12 $O/cpu_scan_pinned.txt
timeout 300 python tools/cpu_baseline_scan.py 3840 2160 32,64,128,256 > $O/cpu_scan_unpinned.txt 2>&1; tail -5 $O/cpu_scan_unpinned.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])" | tee -a $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
