#!/bin/bash
# final round-2 records: rocprofv3 kernel stats + PMC passes of the bench workload, the driver's bench command, the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
export TMPDIR=/tmp
P=$O/prof; mkdir -p $P
W="env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity"
timeout 90 rocprofv3 --kernel-trace --stats -f csv -d $P/trace -o trace -- $W > $P/bench_trace.log 2>&1
timeout 90 rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/pmc1 -o pmc1 -- $W > $P/bench_pmc1.log 2>&1
timeout 90 rocprofv3 -f csv --pmc FETCH_SIZE -d $P/pmc3 -o pmc3 -- $W > $P/bench_pmc3.log 2>&1
timeout 90 rocprofv3 -f csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $P/pmc4 -o pmc4 -- $W > $P/bench_pmc4.log 2>&1
python3 tools/summarize_prof.py $P > $P/summary.txt 2>&1
mkdir -p $P/keep; find $P/trace -name "*kernel_stats.csv" -exec cp {} $P/keep/ \;
rm -rf $P/trace $P/pmc1 $P/pmc3 $P/pmc4
grep -E "gfw_|mean/dispatch|counters" $P/summary.txt | head -30
timeout 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee $O/summary.txt
timeout 100 python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" | tee -a $O/summary.txt
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
python3 -c "
import json
for n in ('driver','default'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['config']['parity_vs_oracle'])"
