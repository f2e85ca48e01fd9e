#!/bin/bash
# full-size decodes of freshly seeded tables, every pack verified in place against the checksum of its source rows
set -e
run() {
  timeout -k 10 400 python bench.py --cpu-seconds 0 --no-exclusive --steps 3 --warmup 1 "$@" > gpurun_out/seed_run.json 2> gpurun_out/seed_run.err
  python - "$*" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/seed_run.json').read().strip().splitlines()[-1])
print(sys.argv[1], '->', d['config']['packs_on_rank0'], 'packs', d['value'], 'GB/s', d['ms_per_step'], 'ms, verified_bit_exact', d['verified_bit_exact'], flush=True)
PY
}
run --seed 7
run --seed 20261112
run --workload numeric --rows 400000000 --seed 3
run --workload varchar --rows 300000000 --seed 5
