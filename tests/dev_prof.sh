#!/bin/bash
# Development aid (GPU box, repo root): kernel trace of a short default-hierarchy bench run.
#   bash tests/dev_prof.sh TAG    -> gpurun_out/TAG/ (kernel trace csv), gpurun_out/TAG.json (bench line)
set -eo pipefail
TAG=$1
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 \
  --secondary 0 --extras 0 --cpu-seconds 1 > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
tail -c 300 $R/gpurun_out/$TAG.json
