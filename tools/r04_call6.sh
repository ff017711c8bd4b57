#!/bin/bash
# round 4, GPU call 6: interop import test, coalescing tests, the whole GPU tier
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_interop.py tests/test_gpu_coalesce.py -m gpu -q -p no:cacheprovider > $O/new_tests.log 2>&1; echo "new tests rc $?" | tee -a $O/summary.txt; tail -30 $O/new_tests.log
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
