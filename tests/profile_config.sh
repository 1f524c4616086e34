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
