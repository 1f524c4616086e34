#!/bin/bash
# Development aid (GPU box, from the repo root): kernel trace of a short bench run -> gpurun_out/<tag>/{stats,bench_stats.json};
# afterwards (anywhere): python tests/launch_timeline.py gpurun_out/<tag>
#   bash tests/dev_trace.sh <tag> [ENV=VAL ...]
set -eo pipefail
TAG=$1
shift || true
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
sha256sum hifir_amd/libhifir_amd.so | cut -d" " -f1 > $OUT/lib.sha256
cat .git_head > $OUT/git_head 2>/dev/null || true
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py \
  --steps 10 --warmup 2 --secondary 0 --extras 0 --cpu-seconds 0.5 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
tail -c 400 $OUT/bench_stats.json
