#!/usr/bin/env python3
"""Reads the file ORR_SCREEN_STAMPS wrote (orr_screen.hip: per workgroup and output tile, s_memtime at tile start, after
the K loop, after the epilogue, and s_memrealtime at the end) and prints where a tile's cycles go."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256, 64, 8)
launch = raw[int(sys.argv[2]) if len(sys.argv) > 2 else -1].copy()
gate = launch.reshape(-1)[-4:].astype(np.int64)      # the last tile slot of the last workgroup doubles as the gate's counters
launch.reshape(-1)[-4:] = 0
if gate[3]:
    print("gate: %d blocks of 32 x 32 seen, %d ran pass 1 (%.1f %%); %d with a lane at thr0; %d lane-0 thr1 <= 0" %
          (gate[3], gate[2], 100.0 * gate[2] / gate[3], gate[1], gate[0]))
s0, s1, s2, rt = (launch[..., k].astype(np.int64) for k in range(4))
e4, e5, e6 = (launch[..., k].astype(np.int64) for k in (4, 5, 6))
valid = s2 > 0
k_loop = (s1 - s0)[valid]
epi = (s2 - s1)[valid]
print("launches in file: %d; tiles stamped in this one: %d over %d workgroups" % (raw.shape[0], valid.sum(), valid.any(axis=1).sum()))
print("K loop   cycles/tile: median %d  p10 %d  p90 %d" % (np.median(k_loop), np.percentile(k_loop, 10), np.percentile(k_loop, 90)))
print("epilogue cycles/tile: median %d  p10 %d  p90 %d" % (np.median(epi), np.percentile(epi, 10), np.percentile(epi, 90)))
first = np.array([np.flatnonzero(v)[0] for v in valid if v.any()])
tot, rts = [], []
for wg in range(256):
    idx = np.flatnonzero(valid[wg])
    if idx.size >= 2:
        tot.append(s2[wg, idx[-1]] - s0[wg, idx[0]])
        rts.append(rt[wg, idx[-1]] - rt[wg, idx[0]])
tot, rts = np.array(tot), np.array(rts)
clk = tot / np.maximum(rts, 1) * 100e6
print("per workgroup: %d tiles, %.0f cycles first start -> last end; clock %.2f GHz (memtime / memrealtime)" %
      (valid.sum(axis=1).max(), np.median(tot), np.median(clk) / 1e9))
print("share of the epilogue: %.1f %%" % (100.0 * epi.sum() / (epi.sum() + k_loop.sum())))
v2 = valid & (e4 > 0) & (e6 > 0)
if v2.any():
    for name, a, b in (("K end -> row constants in registers", s1, e4), ("first block of 32 queries", e4, e5),
                       ("blocks 2..4", e5, e6), ("pass 2 + tail", e6, s2)):
        d = (b - a)[v2]
        print("  %-36s median %6d  p90 %6d" % (name, np.median(d), np.percentile(d, 90)))
s7 = launch[..., 7].astype(np.int64)
v7 = valid & (s7 > 0)
if v7.any():
    print("K loop: start -> tail %d, tail -> end %d (medians)" % (np.median((s7 - s0)[v7]), np.median((s1 - s7)[v7])))
# the first tile of a workgroup includes the cold prologue
seq_k = [(s1 - s0)[:, t][valid[:, t]] for t in range(min(6, valid.shape[1]))]
print("K loop by tile sequence:", [int(np.median(x)) for x in seq_k if x.size])
# real time (s_memrealtime, 100 MHz): when the workgroups began and ended relative to the first one
last = np.array([np.flatnonzero(v)[-1] if v.any() else 0 for v in valid])
wg = np.flatnonzero(valid.any(axis=1))
clk_med = float(np.median(clk))
t_begin = np.array([rt[w, first_i] - (s2[w, first_i] - s0[w, first_i]) / clk_med * 100e6
                    for w, first_i in zip(wg, [np.flatnonzero(valid[w])[0] for w in wg])])
t_end = np.array([rt[w, last[w]] for w in wg]).astype(np.float64)
t0 = t_begin.min()
print("real time, us from the first workgroup's first tile: begins median %.1f p90 %.1f max %.1f; ends min %.1f median %.1f max %.1f" %
      tuple(x / 100.0 for x in (np.median(t_begin - t0), np.percentile(t_begin - t0, 90), (t_begin - t0).max(),
                                (t_end - t0).min(), np.median(t_end - t0), (t_end - t0).max())))
tiles_of = valid.sum(axis=1)[wg]
print("tiles per workgroup: min %d median %d max %d" % (tiles_of.min(), np.median(tiles_of), tiles_of.max()))
