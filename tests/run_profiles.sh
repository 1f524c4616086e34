#!/bin/bash
# Round-end measurement recipe (run on the GPU box through gpurun from the repo root):
#   bash tests/run_profiles.sh r02
# writes gpurun_out/<tag>/...; afterwards (anywhere): python tests/prof_summarize.py gpurun_out/<tag> profiles/<tag>
# Counter passes are separate from the kernel trace (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; --pmc is never combined with other trace domains).
set -eo pipefail
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
# (what ran: the summaries are stamped with THIS, not with whatever library lies in the tree when they are made)
sha256sum hifir_amd/libhifir_amd.so | cut -d" " -f1 > $OUT/lib.sha256
cat .git_head > $OUT/git_head 2>/dev/null || true
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py \
  > $OUT/bench_stats.json 2> $OUT/bench_stats.err
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python3 $R/bench.py \
  --steps 3 --warmup 1 --secondary 0 --extras 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python3 $R/bench.py \
  --steps 3 --warmup 1 --secondary 0 --extras 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
tail -c 600 $OUT/bench_plain.json
