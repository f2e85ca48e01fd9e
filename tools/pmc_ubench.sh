#!/bin/bash
# TCP / TA counters per kernel of a microbenchmark binary: tools/pmc_ubench.sh <tag> "<counters>" <binary> [args]
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
from collections import defaultdict, OrderedDict
acc=OrderedDict(); n=defaultdict(int)
for f in glob.glob(sys.argv[1]+'/sq/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        acc.setdefault(k, defaultdict(float))[r['Counter_Name']]+=float(r['Counter_Value'])
        n[(k,r['Counter_Name'])]+=1
for k,c in acc.items():
    print(k, ' '.join('%s=%.4g'%(a,b/max(n[(k,a)],1)) for a,b in sorted(c.items())))
PY
