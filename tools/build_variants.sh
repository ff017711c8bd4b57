#!/bin/bash
# A/B builds of the hot translation unit (u16, bilinear, C2 instantiation only): variants/libgfwarp_<name>.so (benchmarking only).
# usage: tools/build_variants.sh name:-DFOO=1,-DBAR=2 ...      (library selected at run time with GFW_LIBRARY=...)
#        GFW_VARIANT_TAPS=8 (or 4) tools/build_variants.sh ...  the u16 Lanczos4 / bicubic translation unit instead (always whole)
#        GFW_VARIANT_FULL=1 tools/build_variants.sh ...       whole u16 bilinear translation unit (generic-model instantiations
#                                                             included: bench.py --digital ...; ~2 min per variant)
set -e
cd /root/repo
mkdir -p build/variants variants
HIPCC=/opt/rocm/bin/hipcc
FLAGS="$GFW_VARIANT_EXTRA --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Wno-pass-failed -Iinclude -DGFW_FRAME_KIND=2 -DGFW_FRAME_TAPS=${GFW_VARIANT_TAPS:-2} -DGFW_HOT_ONLY=$([ -n "$GFW_VARIANT_FULL$GFW_VARIANT_TAPS" ] && echo 0 || echo 1)"
[ build/variants/stubs.o -nt tools/variant_stubs.cpp ] || $HIPCC --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Iinclude -c tools/variant_stubs.cpp -o build/variants/stubs.o
build() { name=$1; shift
  $HIPCC $FLAGS "$@" -c gyroflow_amd/csrc/gfw_frame.hip -o build/variants/frame_$name.o 2>build/variants/frame_$name.log || { cat build/variants/frame_$name.log; exit 1; }
  $HIPCC --offload-arch=gfx950 -fPIC -shared build/gfwarp/gfw_api.o build/gfwarp/gfw_kernels.o build/gfwarp/gfw_matrices.o build/variants/stubs.o build/variants/frame_$name.o -o variants/libgfwarp_$name.so
  echo "built $name ($(stat -c %s variants/libgfwarp_$name.so) bytes)"; }
for spec in "$@"; do name=${spec%%:*}; defs=""; [[ "$spec" == *:* ]] && defs=${spec#*:}; build $name $(echo $defs | tr ',' ' ') & done
wait
