#!/bin/bash
# the headline bench for several scratch budgets (smaller budget = more, smaller groups)
set -e
for g in 16 6 3 1.5 0.75; do
  timeout -k 10 300 python bench.py --scratch-gib $g --cpu-seconds 0 --no-exclusive > gpurun_out/budget_$g.json 2> gpurun_out/budget_$g.err
  python - "$g" <<PY
import json, sys
d = json.loads(open("gpurun_out/budget_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("scratch budget GiB", sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], flush=True)
PY
done
