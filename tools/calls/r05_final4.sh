# the certificate sweep by itself, with its output
timeout 600 python3 -m pytest tests/test_gpu_pass1_sweep.py -x -q 2>&1 | tail -40 | cut -c1-600 | tee -a $O/summary.txt
