#!/bin/bash
# round 3, first GPU call: the queued north-star A/B set of round 2 (pruned to the variants that can still change the design)
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh r03a base "base:--streams 2" "base:--streams 4" prio0 atan_tab "atan_tab:--streams 2" "atan_w7:--grid 1792" "atan_w8:--grid 2048" ldsmat "bl_tile:--grid 1280" bl_tile pin atan_pin base \
    "gen3:--digital gopro_superview --steps 60" "gen4:--digital gopro_superview --steps 60" \
    "gen6:--digital gopro_superview --steps 60" "gen2:--digital gopro_superview --steps 60"
