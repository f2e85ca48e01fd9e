#!/bin/bash
# the headline table with the lane + long walkers (built-in choice) against a wave per stream, twice each on one box
set -e
for rep in 1 2; do for lw in 0 1000000; do
  IXR_LANE_WALK_FROM=$lw timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 5 --warmup 2 --no-exclusive > gpurun_out/ab2_${lw}_$rep.json 2> gpurun_out/ab2.err
done; done
python - <<'PY'
import json
for lw in (0,1000000):
  v=[]
  for rep in (1,2):
    j=json.loads(open("gpurun_out/ab2_%d_%d.json"%(lw,rep)).read().strip().splitlines()[-1])
    v.append((j["value"],j["ms_per_step"],[round(k["avg_ms"],3) for k in j["kernels"] if "parse" in k["name"]][0]))
  print("lanes from",lw,v)
PY
