#!/bin/bash
# usage: tools_ab.sh <rows> <lib1> <lib2> ...   -- prints per-kernel avg ms for each build of the library
rows=$1; shift
for lib in "$@"; do
  IXR_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --rows $rows --steps 5 --warmup 2 --cpu-seconds 0 --no-verify > gpurun_out/ab.json 2> gpurun_out/ab.log || { echo "$lib FAILED"; tail -3 gpurun_out/ab.log; continue; }
  python - "$lib" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(sys.argv[1], "step %.3f ms |"%d["ms_per_step"], " ".join("%s=%.3f"%(k["name"],k["avg_ms"]) for k in d["kernels"][:9]))
PY
done
