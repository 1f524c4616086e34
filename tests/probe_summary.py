"""Development aid: summarises the k_trsv_band timestamp probe (library built with `make PROBE=1`, dump written to
$HIFIR_AMD_PROBE_OUT when the handle is closed; wall_clock64 = 100 MHz) per launch and joins it with a rocprofv3
kernel trace of the same program.  Usage: probe_summary.py DUMP TRACE_DIR NODES_PER_APPLY OUT.json"""
import csv, glob, json, sys
import numpy as np
ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256, 16, 16).astype(np.float64)
trace = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)
napply_nodes = int(sys.argv[3])
rows = [r for r in csv.DictReader(open(trace[0])) if "hifamd" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg = rows[-napply_nodes:]
out = []
for pid in range(ts.shape[0]):
    a = ts[pid]
    valid = a[:, :, 0] > 0
    if not valid.any():
        continue
    wgs = np.where(valid.any(axis=1))[0]
    t0 = a[:, :, 0][valid].min()
    ent = (a[:, :, 0][valid] - t0) * 0.01
    syn = ((a[:, :, 2] - a[:, :, 0])[valid]) * 0.01
    ex = a[:, :, 3]
    vex = valid & (ex > 0)
    wrk = ((a[:, :, 3] - a[:, :, 2])[vex]) * 0.01
    span = (ex[vex].max() - t0) * 0.01 if vex.any() else -1
    r = seg[pid]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append(dict(id=pid, name=r["Kernel_Name"].split("(")[0][-30:], wgs=int(len(wgs)), rocprof_us=round(dur, 2), span_us=round(span, 2),
                    entry_max=round(float(ent.max()), 2), sync_med=round(float(np.median(syn)), 2), work_med=round(float(np.median(wrk)), 2) if vex.any() else -1,
                    work_max=round(float(wrk.max()), 2) if vex.any() else -1))
json.dump(out, open(sys.argv[4], "w"))
for o in out[:60]:
    print(o)
