#!/bin/bash
# a varchar table of 24 416 packs: lane + long walkers (the built-in choice from 20 000 string packs) against a wave per stream
set -e
for rep in 1 2; do for lw in 0 1000000; do
  IXR_LANE_WALK_FROM=$lw timeout -k 10 300 python bench.py --workload varchar --rows 400000000 --cpu-seconds 0 --steps 5 --warmup 2 --no-exclusive > gpurun_out/ab3_${lw}_$rep.json 2> gpurun_out/ab3.err
done; done
python - <<'PY'
import json
for lw in (0,1000000):
  v=[]
  for rep in (1,2):
    j=json.loads(open("gpurun_out/ab3_%d_%d.json"%(lw,rep)).read().strip().splitlines()[-1])
    v.append((j["value"],j["ms_per_step"],[round(k["avg_ms"],3) for k in j["kernels"] if "parse" in k["name"]][0]))
  print("lanes from",lw,v)
PY
