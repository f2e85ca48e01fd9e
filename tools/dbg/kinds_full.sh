#!/bin/bash
# full per-kernel breakdown (kernels alone) for the given synthetic column kinds, 1024 packs each
set -e
for k in "$@"; do
  timeout -k 10 120 python bench.py --rows 67108864 --steps 5 --warmup 2 --cpu-seconds 0 --kinds $k > gpurun_out/kindf_$k.json 2> gpurun_out/kindf_$k.err
  python - "$k" <<'PY'
import json,sys
k=sys.argv[1]
d=json.loads(open('gpurun_out/kindf_%s.json'%k).read().strip().splitlines()[-1])
print("kind",k,"value",d['value'],"ms",d['ms_per_step'],"ratio",d['compression_ratio'],"decoded GB",d['config']['decoded_GB'])
for e in d['kernels']: print("   %-24s %8.4f ms  alone %s  %8.1f GB/s alg" % (e['name'],e['avg_ms'],e.get('avg_ms_alone'),e['alg_GBps']))
PY
done
