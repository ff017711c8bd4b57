#!/bin/bash
# A/B runs queued for the first GPU call of the next round (everything below builds in seconds and is bit-exact by construction
# or host-checked; none of it has been timed).  Run on the CPU box:   bash tools/next_round_ab.sh build
# then:   gpurun --timeout 300 -- 'bash tools/next_round_ab.sh run'
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  rm -f variants/*
  bash tools/build_variants.sh "base:" "prio0:-DGFW_PRIO_MODE=0" \
      "atan_tab:-DGFW_ATAN_TABLE=1" "ck30:-DGFW_XCD_CHUNK=30" "atan_ck30:-DGFW_ATAN_TABLE=1,-DGFW_XCD_CHUNK=30" \
      "atan_w7:-DGFW_ATAN_TABLE=1,-DGFW_WAVES_PER_EU=7" "atan_w8:-DGFW_ATAN_TABLE=1,-DGFW_WAVES_PER_EU=8" "ldsmat:-DGFW_LDS_MATRICES=1" "bl_tile:-DGFW_LUT_TILE=2" "age3:-DGFW_PRIO_AGE_ROWS=3" "age6:-DGFW_PRIO_AGE_ROWS=6" "pin:-DGFW_PIN_LENS=1" "atan_pin:-DGFW_ATAN_TABLE=1,-DGFW_PIN_LENS=1" "tl:-DGFW_TIMELINE=1" "tl_atan:-DGFW_TIMELINE=1,-DGFW_ATAN_TABLE=1"
  # register budget of the generic-model instantiations (digital lenses, refraction, IBIS, non-fisheye lenses): whole translation unit
  GFW_VARIANT_FULL=1 bash tools/build_variants.sh "gen3:-DGFW_GENERIC_WAVES_PER_EU=3" "gen4:-DGFW_GENERIC_WAVES_PER_EU=4" \
      "gen6:-DGFW_GENERIC_WAVES_PER_EU=6" "gen2:-DGFW_GENERIC_WAVES_PER_EU=2"
  exit 0
fi
bash tools/gpu_ab.sh r03a base "base:--streams 2" "base:--streams 4" prio0 "prio0:--streams 2" atan_tab "atan_tab:--streams 2" "atan_w7:--grid 1792" "atan_w8:--grid 2048" ldsmat "bl_tile:--grid 1280" bl_tile age3 age6 pin atan_pin ck30 atan_ck30 tl "tl:--streams 2" tl_atan base \
    "gen3:--digital gopro_superview --steps 60" "gen4:--digital gopro_superview --steps 60" \
    "gen6:--digital gopro_superview --steps 60" "gen2:--digital gopro_superview --steps 60"

# ---- staged fused paths (background mode 3, Sony mesh): validate, then promote --------------------------------------------------
# On the CPU box:   GFW_STAGED_FUSED=1 python -c "import __graft_entry__ as g; g.build_gfwarp(force=True)"
# GPU:              gpurun --timeout 300 -- 'python -m pytest tests -m gpu_staged -q'
# Afterwards rebuild the shipped library:   python -c "import __graft_entry__ as g; g.build_gfwarp(force=True)"
# Promotion = drop the GFW_STAGED_FUSED gates, give the two extras their own instantiation (they cost the generic one ~700 B of
# scratch: tools/kernel_resources.py), move the tests to -m gpu.
# Reference-OpenCL second opinion for the other eight lens models: `python -m pytest tests/test_staged_ref_opencl_models.py -m gpu_staged -s`
# prints the agreement per model; put the measured numbers into the assertions and move the file to -m gpu.
# Lanczos4 through the LDS tile (GFW_LUT_TILE, written / compiled / never run): first parity, then time.  The tile kernel holds
# 35.6 KB of LDS and 100 VGPRs: four workgroups per CU, so give it --grid 1024 as well.
#   CPU box:  GFW_VARIANT_TAPS=8 bash tools/build_variants.sh "l8_base:" "l8_tile:-DGFW_LUT_TILE=1" "l8_tilef:-DGFW_LUT_TILE=3"
#   GPU:      bash tools/gpu_ab.sh r03l "l8_base:--interp 8 --steps 60" "l8_tile:--interp 8 --steps 60" "l8_tile:--interp 8 --steps 60 --grid 1024" \
#                                       "l8_tilef:--interp 8 --steps 60 --grid 768"        (f32 tile: 51 KB of LDS, three workgroups per CU)
#   (bench.py checks the last frames of the timed region against the oracle: parity_vs_oracle must read bit-exact)
