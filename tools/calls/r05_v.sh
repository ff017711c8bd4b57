# wave-priority feedback: lane-rows per priority level = a GFW_PRIO_SPAN-th of the wave's work (6 since round 3, tuned on a 55 us kernel)
for sp in 6 3 4 8 10 12 16 24; do bench "GFW_JIT_DEFS=GFW_PRIO_SPAN=$sp" --steps 200; done
bench "GFW_JIT_DEFS=GFW_PRIO_SPAN=6" --steps 200
