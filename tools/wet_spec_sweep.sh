#!/bin/bash
# GPU box: wet marching kernel time vs explicit segment-weight lists (WX_WET_SPEC), alpha 1 and 2
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { python $R/bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms', {a:round(b,4) for a,b in k.items()})"; }
echo "default"; run
for a in ${ALPHAS:-2 1}; do
for spec in "${SPECS[@]:-38x1}" ; do echo "alpha=$a spec=$spec"; WX_WET_ALPHA=$a WX_WET_SPEC=$spec run; done; done
echo "default"; run
