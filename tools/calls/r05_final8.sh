# the pass over the written region behind the per-plane kernel, with output rects
timeout 200 python3 -m pytest tests/test_gpu_checksum.py -q -x -k "rects" 2>&1 | tail -25 | cut -c1-300 | tee -a $O/summary.txt
