#!/bin/bash
# GPU box: the dry marching kernel (BASELINE configs[1] stencil) at the north-star size, three runs, + SQ counters
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do python $R/bench.py --workload dry --X 32768 --Y 4096 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],3), d['roofline']['kernels_ms_per_step'])"; done
for S in 32 48 64 96 128; do echo "MAXSEG=$S"; WX_MARCH_MAXSEG=$S python $R/bench.py --workload dry --X 32768 --Y 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],3))"; done
