#!/bin/bash
# the 1 B-row bench for several builds of the library: step time and the named kernels alone
# usage: tools/dbg/libs_full.sh "<kernel name substrings>" lib1.so lib2.so ...
set -e
pat="$1"; shift
for lib in "$@"; do
  IXR_LIB=$PWD/indexr_amd/$lib timeout -k 10 300 python bench.py --cpu-seconds 0 > gpurun_out/lf_$lib.json 2> gpurun_out/lf_$lib.err
  python - "$lib" "$pat" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/lf_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
pats=sys.argv[2].split()
print(sys.argv[1], d['value'], d['ms_per_step'], d['verified_bit_exact'], [(e['name'], e.get('avg_ms_alone')) for e in d['kernels'] if any(p in e['name'] for p in pats)], flush=True)
PY
done
