export TMPDIR=/tmp
mkdir -p gpurun_out/final
GFW_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/final/rccl_1rank.json 2> gpurun_out/final/rccl_1rank.err; tail -c 400 gpurun_out/final/rccl_1rank.json; tail -2 gpurun_out/final/rccl_1rank.err
python bench.py --no-cpu-baseline --width 7680 --height 4320 --resident 16 --steps 100 > gpurun_out/final/c3.json 2>/dev/null
python bench.py --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 > gpurun_out/final/c4_rgbaf.json 2>/dev/null
python bench.py --no-cpu-baseline --fmt GBRAPF32LE --crop --resident 16 > gpurun_out/final/c4_gbrapf32.json 2>/dev/null
python bench.py --no-cpu-baseline --interp 4 > gpurun_out/final/c2_bicubic.json 2>/dev/null
python bench.py --no-cpu-baseline --interp 8 --steps 100 > gpurun_out/final/c2_lanczos4.json 2>/dev/null
python bench.py --no-cpu-baseline --build-matrices > gpurun_out/final/c2_built.json 2>/dev/null
python bench.py --no-cpu-baseline --upload-matrices > gpurun_out/final/c2_upload.json 2>/dev/null
for f in c3 c4_rgbaf c4_gbrapf32 c2_bicubic c2_lanczos4 c2_built c2_upload; do python -c "
import json,sys; d=json.load(open('gpurun_out/final/$f.json')); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'], d['config']['backend'])"; done
