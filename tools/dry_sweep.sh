#!/bin/bash
# GPU box: the dry marching kernel (BASELINE configs[1] stencil) at the north-star size, three runs, + SQ counters
# (tuning environment switches exist only in the -DWX_DEBUG build of the library: make -C 2d-weather-sandbox_amd/csrc debug)
export WXSIM_LIB=${WXSIM_LIB:-$GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc/variants/libwxsim_debug.so}
[ -f "$WXSIM_LIB" ] || make -C $GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc debug
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do python $R/bench.py --workload dry --X 32768 --Y 4096 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],3), d['roofline']['kernels_ms_per_step'])"; done
for S in 32 48 64 96 128; do echo "MAXSEG=$S"; WX_MARCH_MAXSEG=$S python $R/bench.py --workload dry --X 32768 --Y 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],3))"; done
