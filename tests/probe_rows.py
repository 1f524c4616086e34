"""Development aid: per-row timestamps of k_trsv_band (library built with `make PROBE=1`, dump written to
$HIFIR_AMD_PROBE_OUT when the handle is closed; wall_clock64 = 100 MHz): for the given launch numbers (node index
inside one apply) the waves of the slowest workgroup.  Usage: probe_rows.py DUMP LAUNCH [LAUNCH ...]"""
import sys
import numpy as np
ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256, 16, 16).astype(np.float64)
ids = [int(a) for a in sys.argv[2:]]
for pid in ids:
    a = ts[pid]
    valid = a[:, :, 0] > 0
    t0 = a[:, :, 0][valid].min()
    ex = np.where(valid, a[:, :, 3], 0)
    wg = int(np.argmax(ex.max(axis=1)))
    print(f"== launch {pid}: wgs={int(valid.any(axis=1).sum())} span={(ex.max()-t0)*0.01:.2f} us; slowest wg {wg}")
    for w in range(16):
        r = a[wg, w]
        if r[0] == 0:
            continue
        f = lambda v: (v - t0) * 0.01 if v > 0 else -1
        s = f"  wave {w:2d}: entry {f(r[0]):5.2f} sync {f(r[2]):5.2f} exit {f(r[3]):6.2f} |"
        for k in range(2):
            q = r[4 + 6 * k: 10 + 6 * k]
            if q[0] > 0:
                s += f" row{k}: start {f(q[0]):6.2f} firstgather {f(q[1]):6.2f} done {f(q[2]):6.2f} flag {f(q[3]):6.2f} nnz {int(q[4]):3d} spins {int(q[5]):4d} |"
        print(s)
    # aggregate over all waves of the launch: time from row start to done per nnz
    A = a[:, :, 4][valid]; C = a[:, :, 6][valid]; n = a[:, :, 8][valid]; B = a[:, :, 5][valid]
    ok = (A > 0) & (C > 0)
    d = (C[ok] - A[ok]) * 0.01
    print(f"  all waves row0: n={ok.sum()} nnz med {np.median(n[ok]):.0f}; start->done med {np.median(d):.2f} p90 {np.percentile(d,90):.2f} max {d.max():.2f}; start->firstgather med {np.median((B[ok]-A[ok])*0.01):.2f}")
