#!/bin/bash
# GPU box: wet marching kernel time vs launch shape (segments per strip = rounds x resident waves / strips; cost weight alpha of free-air rows)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { python $R/bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-pmc --no-north-star $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms', {a:round(b,4) for a,b in k.items()})"; }
for r in ${ROUNDS_LIST:-1 2 3 4 5 6}; do for a in ${ALPHA_LIST:-2}; do echo "rounds=$r alpha=$a"; WX_WET_ROUNDS=$r WX_WET_ALPHA=$a run; done; done
