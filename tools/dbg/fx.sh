for v in ${FXV:-base fx1 fx2 fx3 fx4}; do
  IXR_LIB=$PWD/indexr_amd/libixrgpu_$v.so timeout -k 10 120 python bench.py --rows 67108864 --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --kinds 10 > gpurun_out/fx_$v.json 2> gpurun_out/fx_$v.err
  python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/fx_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], [(k['name'],k['avg_ms']) for k in d['kernels'] if k['name'] in ('k_str_fused','k_str_bases','k_str_index')])
PY
done
