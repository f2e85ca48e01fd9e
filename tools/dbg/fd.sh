for v in ${FXV}; do
 for k in 10 11; do
  IXR_LIB=$PWD/indexr_amd/libixrgpu$v.so timeout -k 10 120 python bench.py --rows 67108864 --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --kinds $k > gpurun_out/fd.json 2> gpurun_out/fd.err
  python - "$v" $k <<'PY'
import json,sys
d=json.loads(open('gpurun_out/fd.json').read().strip().splitlines()[-1])
print(sys.argv[1] or 'base', 'kind', sys.argv[2], [(k['name'],k['avg_ms']) for k in d['kernels'] if k['name'] in ('k_front_decode','k_front_summary','k_str_fused')])
PY
 done
done
