#!/bin/bash
# round 6, call r: the LDS-staged tap probe (tools/lds_taps_probe.hip): Lanczos4 / bicubic / bilinear taps from a window of floats in LDS against today's fetch
O=gpurun_out/r06_r; mkdir -p $O
./tools/lds_taps_probe > $O/lds_taps_probe.txt 2>&1; cat $O/lds_taps_probe.txt
