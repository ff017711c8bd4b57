#!/usr/bin/env python3
"""Summarise a GFW_TIMELINE dump (diagnosis builds of the fused kernel): per-wave start / end on the 100 MHz device clock,
phase clocks, HW_ID / XCC_ID.  usage: analyze_timeline.py file.bin [n_wg]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
n = int(sys.argv[2]) * 4 if len(sys.argv) > 2 else 1536 * 4
a = a[:n]
start, end = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
t0 = start.min()
s_us, e_us = (start - t0) / 100.0, (end - t0) / 100.0
life = e_us - s_us
print("waves %d   kernel span %.1f us   start: min %.1f max %.1f mean %.1f   end: min %.1f max %.1f mean %.1f" % (n, e_us.max(), s_us.min(), s_us.max(), s_us.mean(), e_us.min(), e_us.max(), e_us.mean()))
setup = (a[:, 7].astype(np.int64) >> 32) / 100.0
print("set-up before the tile walk, us: mean %.2f  p50 %.2f  p95 %.2f  max %.2f" % (setup.mean(), *np.percentile(setup, [50, 95]), setup.max()))
print("wave lifetime us: mean %.1f  p5 %.1f  p50 %.1f  p95 %.1f max %.1f" % (life.mean(), *np.percentile(life, [5, 50, 95]), life.max()))
units = a[:, 4].astype(np.int64)
print("units(rows)/wave: min %d max %d mean %.2f;  us per row: mean %.2f" % (units.min(), units.max(), units.mean(), (life / np.maximum(units, 1)).mean()))
p1, p3 = a[:, 2].astype(np.float64), a[:, 3].astype(np.float64)
print("phase clocks (s_memtime ticks): phase1+flush %.3g  phase3 %.3g  ratio p1/(p1+p3) %.3f" % (p1.sum(), p3.sum(), p1.sum() / (p1.sum() + p3.sum())))
hw = a[:, 5].astype(np.int64); xcc = a[:, 6].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
print("XCC ids seen:", sorted(set(xcc.tolist())), " blockIdx&7 == xcc for %.1f %% of waves" % (100.0 * np.mean(((a[:, 7].astype(np.int64) & 0xffffffff) & 7) == xcc)))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  xcc %d: waves %4d  start mean %.1f  end mean %.1f max %.1f  life mean %.1f  rows %d" % (x, m.sum(), s_us[m].mean(), e_us[m].mean(), e_us[m].max(), life[m].mean(), units[m].sum()))
key = xcc * 1000 + se * 100 + sh * 50 + cu
ks, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu): %d   waves per CU: min %d max %d" % (len(ks), cnt.min(), cnt.max()))
ends = np.array([e_us[key == k].max() for k in ks]); rows = np.array([units[key == k].sum() for k in ks])
print("per-CU last end us: min %.1f p50 %.1f max %.1f;   rows per CU: min %d max %d" % (ends.min(), np.median(ends), ends.max(), rows.min(), rows.max()))
# occupancy over time
ts = np.linspace(0, e_us.max(), 21)
occ = [(np.sum((s_us <= t) & (e_us > t))) / 1024.0 for t in ts]
print("resident waves per SIMD over time:", " ".join("%.1f" % o for o in occ))
# how level the launch is: wave-slot time that does no work before the last wave ends, where the waves end, and the same per XCD
span = e_us.max() - s_us.min()
print("busy fraction of the launch: %.4f   (idle before start %.4f, idle after end %.4f)" % (
    life.sum() / (n * span), (s_us - s_us.min()).sum() / (n * span), (e_us.max() - e_us).sum() / (n * span)))
print("wave end us percentiles 1/5/25/50/75/95/99/100: " + " ".join("%.1f" % v for v in np.percentile(e_us, [1, 5, 25, 50, 75, 95, 99, 100])))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  xcc %d: end p5 %.1f p50 %.1f max %.1f   idle-after-end share %.4f" % (x, *np.percentile(e_us[m], [5, 50]), e_us[m].max(), (e_us.max() - e_us[m]).sum() / (m.sum() * span)))
simd_key = key * 4 + simd
sk = np.unique(simd_key)
send = np.array([e_us[simd_key == k].max() for k in sk])
print("per-SIMD last end us: p5 %.1f p50 %.1f p95 %.1f max %.1f" % (*np.percentile(send, [5, 50, 95]), send.max()))
per_row = life / np.maximum(units, 1)
print("us per lane-row by wave: p5 %.3f p50 %.3f p95 %.3f" % tuple(np.percentile(per_row, [5, 50, 95])))
