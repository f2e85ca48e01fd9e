# how long the lane walker's chains are, per column kind (the launch waits for the longest chain): tools/dbg/lane_chain.sh <kinds...>
for k in "$@"; do
  IXR_LANE_WALK_FROM=1 timeout -k 10 120 python bench.py --rows 67108864 --steps 5 --warmup 2 --cpu-seconds 0 --no-exclusive --kinds $k > gpurun_out/lane_$k.json 2> gpurun_out/lane_$k.err
  python - $k <<'PY'
import json,sys
d=json.loads(open('gpurun_out/lane_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('kind', sys.argv[1], [(k['name'],k['avg_ms']) for k in d['kernels'] if k['name']=='k_parse'])
PY
done
