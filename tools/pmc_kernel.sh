#!/bin/bash
# SQ counters of one experiment workload: tools/pmc_kernel.sh <tag> <bench args...>   (runs on the GPU box)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY} --output-format csv -d $O/sq -- python3 $ROOT/bench.py --cpu-seconds 0 --steps 2 --warmup 1 --no-verify "$@" > /dev/null 2> $O/sq.log
python3 - $O <<'PY'
import csv,glob,sys,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for f in glob.glob(sys.argv[1]+'/sq/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.match(r'(?:void )?(?:ixr::)?(k_\w+)(<[^>]*>)?', r['Kernel_Name'])
        if not m: continue
        k=m.group(1)+(m.group(2) or '')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        n[(k,r['Counter_Name'])]+=1
for k,c in acc.items():
    print(k, ' '.join('%s=%.4g'%(a,b/max(n[(k,a)],1)) for a,b in sorted(c.items())))
PY
