#!/bin/bash
# after tools/profile.sh <tag>, numeric, varchar on the GPU box: keep the newest files of each tag locally, write the traffic
# summaries and copy the round's record.   usage: tools/dbg/collect_profiles.sh <tag> <round dir, e.g. profiles/r3>
set -e
TAG=$1; R=$2
for w in numeric varchar $TAG; do for sub in stats pmc_fetch pmc_write; do d=gpurun_out/prof_$w/$sub/vm; newest=$(ls -t $d/*_agent_info.csv | head -1 | xargs basename | cut -d_ -f1); for f in $d/*; do b=$(basename $f | cut -d_ -f1); [ "$b" != "$newest" ] && rm -f $f; done; done; done
python tools/pmc_traffic.py gpurun_out/prof_$TAG profiles/pmc_traffic.json > /dev/null
python tools/pmc_traffic.py gpurun_out/prof_numeric profiles/pmc_traffic_numeric.json > /dev/null
python tools/pmc_traffic.py gpurun_out/prof_varchar profiles/pmc_traffic_varchar.json > /dev/null
cp profiles/pmc_traffic.json $R/${TAG}_pmc_traffic.json; cp profiles/pmc_traffic_numeric.json $R/cfg2_pmc_traffic.json; cp profiles/pmc_traffic_varchar.json $R/cfg3_pmc_traffic.json
cp gpurun_out/prof_$TAG/stats/vm/*_kernel_stats.csv $R/${TAG}_kernel_stats.csv
cp gpurun_out/prof_numeric/stats/vm/*_kernel_stats.csv $R/cfg2_kernel_stats.csv
cp gpurun_out/prof_varchar/stats/vm/*_kernel_stats.csv $R/cfg3_kernel_stats.csv
cp gpurun_out/prof_$TAG/bench_stats.json $R/${TAG}_bench_under_rocprof.json
cp gpurun_out/prof_numeric/bench_stats.json $R/cfg2_numeric_100m.json
cp gpurun_out/prof_varchar/bench_stats.json $R/cfg3_varchar_100m.json
