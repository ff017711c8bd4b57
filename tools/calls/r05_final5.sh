# the driver's own command: the whole GPU suite, serially, one process
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -60 | cut -c1-400 | tee -a $O/summary.txt
