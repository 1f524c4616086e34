"""Development aid: per-batch timeline of the first row of every wave of k_trsv_band_p (library built with
`make PROBE=2`, dump written to $HIFIR_AMD_PROBE_OUT when the handle is closed; wall_clock64 = 100 MHz): for the given
launch numbers, issue->accumulated latency of the leading full batches (eight gathers each) and the gap between
batches.  Usage: probe_batches.py DUMP LAUNCH [LAUNCH ...]"""
import sys

import numpy as np

ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256, 16, 16).astype(np.float64)
for pid in [int(a) for a in sys.argv[2:]]:
    a = ts[pid]
    valid = a[:, :, 0] > 0
    if not valid.any():
        print(f"== launch {pid}: not recorded (not a k_trsv_band_p launch)")
        continue
    t0 = a[:, :, 0][valid].min()
    lat, gap, first = [], [], []
    nb = []
    for wg in range(256):
        for w in range(16):
            r = a[wg, w]
            if r[0] == 0 or r[4] == 0:
                continue
            k = 0
            while k < 6 and r[4 + 2 * k] > 0 and r[5 + 2 * k] > 0:
                lat.append((r[5 + 2 * k] - r[4 + 2 * k]) * 0.01)
                if k:
                    gap.append((r[4 + 2 * k] - r[3 + 2 * k]) * 0.01)
                else:
                    first.append((r[4] - t0) * 0.01)
                k += 1
            nb.append(k)
    if not lat:
        print(f"== launch {pid}: no heavy rows")
        continue
    lat, gap, first = np.array(lat), np.array(gap if gap else [0.0]), np.array(first)
    print(f"== launch {pid}: {int(valid.any(axis=1).sum())} workgroups, {len(nb)} first rows with leading batches "
          f"(median {np.median(nb):.0f} per row)")
    print(f"   first batch issued {np.median(first):.2f} us after the kernel's first wave (p90 {np.percentile(first, 90):.2f})")
    print(f"   issue -> all eight accumulated: median {np.median(lat):.2f} us, p10 {np.percentile(lat, 10):.2f}, "
          f"p90 {np.percentile(lat, 90):.2f}, max {lat.max():.2f}")
    print(f"   accumulated -> next batch issued (poll, broadcasts, address arithmetic): median {np.median(gap):.2f} us, "
          f"p90 {np.percentile(gap, 90):.2f}, max {gap.max():.2f}")
