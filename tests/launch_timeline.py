"""Per-launch timeline of ONE apply from a rocprofv3 kernel trace of bench.py (tests/run_profiles.sh):
  python tests/launch_timeline.py gpurun_out/<tag> [out.txt]
One line per launch of the apply: index, level, stage (the library's own launch map), kernel, workgroups, threads per
workgroup, LDS bytes, median duration over the profiled applies and the gap to the previous launch."""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout


def periodic_windows(sig, L_):
    best, cur_start = (0, 0), None
    for i in range(len(sig) - L_):
        if sig[i] == sig[i + L_]:
            if cur_start is None:
                cur_start = i
            if i + 1 - cur_start > best[1] - best[0]:
                best = (cur_start, i + 1)
        else:
            cur_start = None
    p0, p1 = best[0], best[1] + L_
    res, e = [], p1
    while e - L_ >= p0:
        res.append((e - L_, e - 1))
        e -= L_
    res.reverse()
    return res


kt = glob.glob(os.path.join(src, "stats/*/*kernel_trace.csv"))[0]
line = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
L_ = int(line["config"]["launches_per_apply"])
lmap = line["roofline"]["launch_map"]
rows = sorted((r for r in csv.DictReader(open(kt)) if "hifamd" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
wins = periodic_windows([(r["Kernel_Name"], r["Grid_Size_X"]) for r in rows], L_)
stage_name = {1: "S1/L", 2: "LDU1", 3: "E", 4: "tail", 5: "F", 6: "LDU2", 7: "out"}
for f in ("lib.sha256", "git_head"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        print(f"# {f}: {open(p).read().strip()}", file=out)
print(f"# applies in the trace: {len(wins)}; launches per apply: {L_}; durations: median over the applies, microseconds", file=out)
print("# idx level stage  kernel                                   wgs  thr   lds_B   us     gap_us  cum_us", file=out)
cum = 0.0
for j in range(L_):
    ds, gs = [], []
    for (a, b) in wins:
        q = rows[a + j]
        ds.append((int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3)
        if j > 0:
            gs.append((int(q["Start_Timestamp"]) - int(rows[a + j - 1]["End_Timestamp"])) / 1e3)
    q = rows[wins[0][0] + j]
    kn = q["Kernel_Name"].replace("void hifamd::", "").replace("hifamd::", "")
    kn = kn.split("(")[0].replace(" ", "")
    wg = int(q["Workgroup_Size_X"])
    d = statistics.median(ds)
    g = statistics.median(gs) if gs else 0.0
    cum += d
    print(f"{j:4d} {lmap[j] // 16:3d}  {stage_name.get(lmap[j] % 16, '?'):5s}  {kn[:40]:40s} {int(q['Grid_Size_X']) // wg:5d} {wg:5d} {int(q['LDS_Block_Size']):7d} {d:7.1f} {g:7.2f} {cum:8.1f}", file=out)
