#!/bin/bash
# timing ablations of gfw_hot_kernel (variant 32 + bits: 1 no first pass, 2 no taps, 4 no exact resolve, 8 accept everything, 16 no table load, 32 uniform matrix row, 64 no luma store)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
for v in ${VARIANTS:-47 63 79 111 95 143 159}; do
    timeout 200 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --variant $v --resident 16 > gpurun_out/r02e/v${v}.json 2> gpurun_out/r02e/v${v}.err
    python3 -c "import json; d=json.load(open('gpurun_out/r02e/v${v}.json')); print('variant', $v, 'ablate', $v-32, 'us/step', round(d['ms_per_step']*1000,2), 'kernel', round(d['roofline']['kernel_ms_per_launch']*1000,2), d['config']['backend'])" | tee -a gpurun_out/r02e/summary.txt
done
