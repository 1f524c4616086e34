"""Development aid: summarises a HIFIR_AMD_CSPROBE_OUT dump (make CSPROBE=1): per launch of a component band kernel the
phase boundaries of its workgroups, relative to the launch's first workgroup entry (microseconds).
  python tests/dev_csprobe.py dump.bin [verbose]"""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12)
kind = (a[:, 8] >> np.uint64(32)).astype(np.int64)
blk = (a[:, 8] & np.uint64(0xffffffff)).astype(np.int64)
grid = (a[:, 9] >> np.uint64(32)).astype(np.int64)
nband = (a[:, 9] & np.uint64(0xffffffff)).astype(np.int64)
nsl = a[:, 10].astype(np.int64)
xcc = a[:, 11].astype(np.int64)
t = a[:, :8].astype(np.float64) / 100.0  # 100 MHz -> us
# launches: order by entry time, cut where (grid, nband, nsl, kind & 28) changes
order = np.argsort(t[:, 0], kind="stable")
key = np.stack([grid, nband, nsl, kind & 28], 1)[order]
cuts = [0] + [i for i in range(1, len(order)) if (key[i] != key[i - 1]).any()] + [len(order)]
print("launches:", len(cuts) - 1, "records:", len(a))
T0 = t[order[0], 0]
prev_end = None
for c in range(len(cuts) - 1):
    idx = order[cuts[c]:cuts[c + 1]]
    k = kind[idx]
    comp = idx[(k & 3) == 1]
    carr = idx[(k & 3) == 2]
    t0 = t[idx, 0].min()
    end = t[idx, 6].max()
    name = ("cd" if (k[0] & 16) else "cs") + ("U" if (k[0] & 4) else "L") + ("s" if (k[0] & 8) else "")
    gap = (t0 - prev_end) if prev_end is not None else 0.0
    prev_end = end
    line = f"{t0 - T0:9.1f} {name:5s} grid={grid[idx[0]]:5d} nband={nband[idx[0]]:5d} nsl={nsl[idx[0]]} gap={gap:5.1f} dur={end - t0:6.1f}"
    if len(comp):
        tc = t[comp]
        ent = tc[:, 0] - t0
        ph = [np.where(tc[:, j] > 0, tc[:, j] - tc[:, 0], np.nan) for j in range(1, 7)]
        def mx(x):
            return np.nanmax(x) if np.isfinite(x).any() else float("nan")
        def md(x):
            return np.nanmedian(x) if np.isfinite(x).any() else float("nan")
        line += f" | entry max {ent.max():5.1f} | med/max since entry: ph0 {md(ph[1]):4.1f}/{mx(ph[1]):4.1f} 1a {md(ph[2]):4.1f}/{mx(ph[2]):4.1f} 1b {md(ph[4]):4.1f}/{mx(ph[4]):4.1f} end {md(ph[5]):4.1f}/{mx(ph[5]):4.1f}"
    if len(carr):
        tc = t[carr]
        line += f" | carried n={len(carr)} entry max {(tc[:, 0] - t0).max():5.1f} end max {(tc[:, 6] - t0).max():5.1f}"
    print(line)

# optional: per-workgroup detail of one launch:  python tests/dev_csprobe.py dump.bin detail GRID NBAND
if len(sys.argv) > 4 and sys.argv[2] == "detail":
    G, NB = int(sys.argv[3]), int(sys.argv[4])
    for c in range(len(cuts) - 1):
        idx = order[cuts[c]:cuts[c + 1]]
        if grid[idx[0]] != G or nband[idx[0]] != NB:
            continue
        t0 = t[idx, 0].min()
        print("launch at %.1f: per workgroup (block, xcc): entry, then phase stamps relative to entry" % (t0 - T0))
        rows = sorted(idx, key=lambda i: -(t[i, 6] - t[i, 0]))
        for i in rows[:12] + rows[-4:]:
            st = ["%5.1f" % (t[i, j] - t[i, 0]) if t[i, j] > 0 else "  -  " for j in range(1, 7)]
            print("  blk %4d xcc %d kind %2d entry %5.1f | %s" % (blk[i], xcc[i] & 7, kind[i], t[i, 0] - t0, " ".join(st)))
        break
