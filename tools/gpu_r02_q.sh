#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh r02q dyn "dyn:--tune-rows 50" "dyn:--tune-rows 100" "dyn:--tune-rows -1" "dyn:--variant 4" dyn_ck30 dyn_ru4 dyn_ru1 tl_dyn
O=gpurun_out/r02q
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee -a $O/summary.txt
python3 -c "import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['config']['parity_vs_oracle'])"
