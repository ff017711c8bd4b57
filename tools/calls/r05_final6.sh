# the driver's bench command on the final tree (the traffic file now carries this library's kernel_source_id)
timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json | tee -a $O/summary.txt
