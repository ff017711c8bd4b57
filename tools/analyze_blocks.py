#!/usr/bin/env python3
"""Where a wave's life goes inside the branch-free lane-row (GFW_JIT_DEFS=GFW_TIMELINE=2 builds, $GFW_TIMELINE_FILE.blocks): per wave the accumulated shader clocks of
[row start .. matrix rows arrived], [.. projection done], [.. luma taps + store], [.. chroma], with the first pass and the whole of phase 3 beside them.
usage: analyze_blocks.py file.bin.blocks [n_wg]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
n = int(sys.argv[2]) * 4 if len(sys.argv) > 2 else 2048 * 4
a = a[:n]
a = a[a[:, 4] > 0]
blk, rows, p1, p3 = a[:, :4], a[:, 4], a[:, 5], a[:, 6]
tot = p1.sum() + p3.sum()
names = ["matrix rows: LDS row read, six fetches, wait", "projection (both pixels, to u, v)", "luma: map, bins, votes, taps, store", "chroma site: bins, vote, two planes' taps, stores"]
print("waves %d, lane-rows per wave %.1f; clocks per lane-row: first pass + queue %.0f, phase 3 %.0f" % (len(a), rows.mean(), (p1 / rows).mean(), (p3 / rows).mean()))
for k in range(4):
    print("  %-52s %7.0f clocks per lane-row  %5.1f %% of the wave's tile time" % (names[k], (blk[:, k] / rows).mean(), 100.0 * blk[:, k].sum() / tot))
print("  %-52s %7.0f clocks per lane-row  %5.1f %%" % ("first pass (nodes, rows, queue resolution)", (p1 / rows).mean(), 100.0 * p1.sum() / tot))
print("  %-52s %7.0f clocks per lane-row  %5.1f %%" % ("rest of phase 3 (loop, fences, priority)", ((p3 - blk.sum(axis=1)) / rows).mean(), 100.0 * (p3.sum() - blk.sum()) / tot))
