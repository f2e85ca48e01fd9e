#!/bin/bash
# SQ counters of any command: tools/pmc_any.sh <tag> "<counters>" <program> [args...]   (runs on the GPU box; the program
# itself must follow, not a shell or env wrapper)
set -e
TAG=$1; shift
PMC=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $O/sq -- "$@" > $O/out.log 2> $O/sq.log
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
for k,c in sorted(acc.items()):
    print(k, ' '.join('%s=%.4g'%(a,b/max(n[(k,a)],1)) for a,b in sorted(c.items())))
PY
