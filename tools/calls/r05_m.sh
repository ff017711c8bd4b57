# smoke(), the driver's bench command, the whole GPU suite on the final tree
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 | cut -c1-600
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
