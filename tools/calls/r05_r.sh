# latency: fewer, longer waits per lane-row.  C2 bilinear with the row's fetches in one cluster (measured neutral at round start, on a kernel with 10 % more instructions); Lanczos4 with more rows in flight at fewer waves
bench "GFW_JIT_DEFS=GFW_ROW_CLUSTER=1" --steps 200
bench "GFW_JIT_DEFS=GFW_ROW_CLUSTER=0" --steps 200
bench "GFW_JIT_DEFS=GFW_ROW_CLUSTER=1" --steps 200
bench "GFW_JIT_DEFS=GFW_ROW_CLUSTER=0" --steps 200
bench "GFW_JIT_WAVES=7 GFW_JIT_DEFS=GFW_ROW_CLUSTER=1" --steps 200
for w in 4 5 6; do for r in 4 8; do bench "GFW_JIT_WAVES=$w GFW_JIT_DEFS=GFW_TAP_ROWS_FORCE=$r" --interp 8 --steps 100; done; done
bench "GFW_JIT_WAVES=5 GFW_JIT_DEFS=GFW_TAP_ROWS_FORCE=4" --interp 4
bench "GFW_JIT_WAVES=4 GFW_JIT_DEFS=GFW_TAP_ROWS_FORCE=4" --interp 4
