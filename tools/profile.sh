#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics plus the two HBM-traffic counter passes of the default bench workload.
# usage: tools/profile.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_$TAG
rm -rf $O   # a tag that is run again must not add its counter files to the old ones
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --cpu-seconds 0 --no-exclusive $*"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_stats.json 2> $O/stats.log
echo "stats pass done"
timeout -s KILL 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 2 --warmup 1 --no-verify > /dev/null 2> $O/pmc_fetch.log
echo "fetch pass done"
timeout -s KILL 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 2 --warmup 1 --no-verify > /dev/null 2> $O/pmc_write.log
echo "write pass done"
# keep only what is small: the per-dispatch counter csv of 1B-row runs is modest (a few thousand dispatches)
find $O -name '*_kernel_trace.csv' -size +20M -delete
ls -la $O/*/*/ | head -30
