#!/bin/bash
# per-kind kernel timings: one bench line per synthetic column kind (experiment helper)
set -e
ROWS=${ROWS:-67108864}
for k in "$@"; do
  timeout -k 10 120 python bench.py --rows $ROWS --steps 5 --warmup 2 --cpu-seconds 0 --kinds $k > gpurun_out/kind_$k.json 2> gpurun_out/kind_$k.err
  python - "$k" <<'PY'
import json,sys
k=sys.argv[1]
d=json.loads(open('gpurun_out/kind_%s.json'%k).read().strip().splitlines()[-1])
print("kind",k,"value",d['value'],"ms",d['ms_per_step'],"ratio",d['compression_ratio'])
for e in d['kernels']: print("   %-18s %8.4f ms  %8.1f GB/s alg" % (e['name'],e['avg_ms'],e['alg_GBps']))
PY
done
