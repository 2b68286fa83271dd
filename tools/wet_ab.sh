#!/bin/bash
# GPU box: interleaved A/B of launch-shape settings (the boxes drift by several % within a call: never compare across time)
# usage: CONFIGS=("ENV=.. ENV=.." "...") REPS=4 . tools/wet_ab.sh
# (tuning environment switches exist only in the -DWX_DEBUG build of the library: make -C 2d-weather-sandbox_amd/csrc debug)
export WXSIM_LIB=${WXSIM_LIB:-$GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc/variants/libwxsim_debug.so}
[ -f "$WXSIM_LIB" ] || make -C $GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc debug
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { python $R/bench.py --steps ${STEPS:-150} --warmup 10 --no-cpu-baseline --no-pmc --no-north-star $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print(round(d['roofline']['avg_launch_ms'],4), end=' ')"; }
export -f run; export R STEPS BENCH_ARGS
for cfg in "${CONFIGS[@]}"; do printf "%-70s" "$cfg"; for i in $(seq ${REPS:-4}); do :; done; echo; done > /dev/null
for i in $(seq ${REPS:-4}); do for cfg in "${CONFIGS[@]}"; do printf "%-75s" "[$i] $cfg: "; env $cfg bash -c run; echo; done; done
