#!/bin/bash
# A/B builds of the hot translation unit (u16, bilinear): libgfwarp_<name>.so under build/variants (benchmarking only)
set -e
cd /root/repo
mkdir -p build/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -DGFW_FRAME_KIND=2 -DGFW_FRAME_TAPS=2"
build() { name=$1; shift; /opt/rocm/bin/hipcc $FLAGS "$@" -c gyroflow_amd/csrc/gfw_frame.hip -o build/variants/frame_$name.o 2>/dev/null; 
  objs=$(ls build/gfwarp/*.o | grep -v gfw_frame_k2_t2.o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs build/variants/frame_$name.o -o build/variants/libgfwarp_$name.so; echo built $name; }
for spec in "$@"; do name=${spec%%:*}; defs=${spec#*:}; build $name $(echo $defs | tr ',' ' ') & done
wait
