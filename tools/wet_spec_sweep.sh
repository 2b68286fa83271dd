#!/bin/bash
# GPU box: wet marching kernel time vs explicit segment-weight lists (WX_WET_SPEC), alpha 1 and 2
# (tuning environment switches exist only in the -DWX_DEBUG build of the library: make -C 2d-weather-sandbox_amd/csrc debug)
export WXSIM_LIB=${WXSIM_LIB:-$GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc/variants/libwxsim_debug.so}
[ -f "$WXSIM_LIB" ] || make -C $GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc debug
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { python $R/bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms', {a:round(b,4) for a,b in k.items()})"; }
echo "default"; run
for a in ${ALPHAS:-2 1}; do
for spec in "${SPECS[@]:-38x1}" ; do echo "alpha=$a spec=$spec"; WX_WET_ALPHA=$a WX_WET_SPEC=$spec run; done; done
echo "default"; run
