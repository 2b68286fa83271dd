#!/bin/bash
# GPU box: the overlapped exchange protocol of one slab (tools/slab_protocol_cost.py, ONE handle for all modes) under the tuning
# switches of the split iterations (debug build only): WX_SPLIT_PRIO = s_setprio level of the edge waves, WX_SPLIT_NOFENCE=1 = no
# release fence in front of an edge wave's arrival in the ordered single launch (timing only: wrong results), WX_SPLIT_HALVE=0 = the
# edge group of the two-launch protocol with full-height segments. Usage: split_variants.sh [X_global Y]
R=$GRAFT_REPO_ROOT
export WXSIM_LIB=${WXSIM_LIB:-$R/2d-weather-sandbox_amd/csrc/variants/libwxsim_debug.so}
[ -f "$WXSIM_LIB" ] || make -C $R/2d-weather-sandbox_amd/csrc debug
XG=${1:-16384}; Y=${2:-2048}
for v in ${VARIANTS:-"PRIO=0,HALVE=1,NOFENCE=0" "PRIO=3,HALVE=1,NOFENCE=0" "PRIO=3,HALVE=0,NOFENCE=0" "PRIO=0,HALVE=0,NOFENCE=0" "PRIO=3,HALVE=1,NOFENCE=1"}; do
  echo "== $v"
  env $(echo $v | sed 's/,/ /g; s/\([A-Z]*\)=/WX_SPLIT_\1=/g') TUNE=${TUNE:-0} timeout 200 python $R/tools/slab_protocol_cost.py $XG $Y 42 ${REPS:-3} 2>&1 | grep -v amdgpu.ids
done
