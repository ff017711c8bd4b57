#!/bin/bash
# round 4, GPU call 25: wide device audit of the derived certificate (r_limit, shifted coordinates), the first-pass tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python3 tools/audit_sweep_r04.py 3000 > $O/sweep.txt 2>&1; echo "sweep rc $?" | tee -a $O/summary.txt; tail -6 $O/sweep.txt | tee -a $O/summary.txt

