#!/bin/bash
# Kernel summary of one secondary BASELINE configuration (GPU box, repo root), stamped with the git head and the
# sha256 of the library that ran:
#   bash tests/profile_config.sh TAG NX MODE NRHS     e.g.  r03_config5 816 kkt 16   |   r03_config4 256 tuned-3d 64
# writes gpurun_out/TAG_kernels.txt (copy it to profiles/).  rocprofv3 --kernel-trace --stats of tests/perf_probe.py.
set -eo pipefail
TAG=$1; NX=$2; MODE=$3; NRHS=$4
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
SHA=$(sha256sum $R/hifir_amd/libhifir_amd.so | cut -d' ' -f1)
HEAD=$(cat $R/.git_head 2>/dev/null || git -C $R rev-parse HEAD 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1100 rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 $R/tests/perf_probe.py $NX $MODE $NRHS \
  > $OUT/probe.log 2> $OUT/probe.err
{
  echo "# BASELINE secondary configuration: tests/perf_probe.py $NX $MODE $NRHS under rocprofv3 --kernel-trace --stats"
  echo "# git_head=$HEAD lib_sha256=$SHA"
  grep -E "reference factorize|import\+upload|first apply|relerr|RESULT" $OUT/probe.log
  echo "# per level, from the LAST apply of the trace attributed with the library's launch map (microseconds; bytes: SURVEY 8(d))"
  python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
lm = None
for ln in open(out + "/probe.log"):
    if ln.startswith("LAUNCHMAP "):
        lm = json.loads(ln[10:])
kt = glob.glob(out + "/*/*kernel_trace.csv")
if lm and kt:
    rows = sorted((r for r in csv.DictReader(open(kt[0])) if "hifamd" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    mp = lm["map"]
    last = rows[-len(mp):]
    grp = {1: "in+LDU", 2: "in+LDU", 3: "E", 4: "dense/tail", 5: "F+LDU+out", 6: "F+LDU+out", 7: "F+LDU+out"}
    t, cnt = {}, {}
    for r, m in zip(last, mp):
        key = (m // 16, grp.get(m % 16, "other"))
        t[key] = t.get(key, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[key] = cnt.get(key, 0) + 1
    tot = sum(t.values())
    for lv in sorted({k[0] for k in t}):
        lb = sum(float(v) for v in lm["level_bytes"].get(str(lv), {}).values())
        us = sum(v for k, v in t.items() if k[0] == lv)
        parts = "  ".join(f"{g} {t[(lv, g)]:.0f} us / {cnt[(lv, g)]} launches" for g in ("in+LDU", "E", "dense/tail", "F+LDU+out") if (lv, g) in t)
        rows_ = lm["level_rows"][lv] if lv < len(lm["level_rows"]) else 0
        print(f"level {lv} ({rows_} rows): {us:9.0f} us  {100 * us / tot:5.1f} %  {lb / 1e9:7.2f} GB  {lb / us / 1e6 if us else 0:6.2f} TB/s   {parts}")
    print(f"one apply: {len(mp)} launches, {tot / 1e3:.3f} ms of kernel time")
PY
  echo "# kernels (all applies of the run: warm-up, parity check and timed repetitions), by total time"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "hifamd" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-58s calls=%6s total=%9.2f ms avg=%8.1f us  %5.1f %%" % (r["Name"].split("(")[0].replace("void hifamd::", "")[:58], r["Calls"],
          float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
} > $R/gpurun_out/${TAG}_kernels.txt
cat $R/gpurun_out/${TAG}_kernels.txt
