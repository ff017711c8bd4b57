#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee $O/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
cat $O/bench_driver.json
