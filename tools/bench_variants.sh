#!/bin/bash
# GPU box: run the bench for every tuning variant of libwxsim and print the per-kernel times. ARGS = extra bench args
R=$GRAFT_REPO_ROOT
for f in $R/2d-weather-sandbox_amd/csrc/variants/libwxsim_*.so; do
  n=$(basename $f .so)
  WXSIM_LIB=$f python $R/bench.py --steps ${STEPS:-60} --warmup 6 --no-cpu-baseline --no-pmc $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('$n', round(d['value']), 'Mcs/s', round(d['ms_per_step'],3), 'ms', {a:round(b,3) for a,b in k.items()})"
done
