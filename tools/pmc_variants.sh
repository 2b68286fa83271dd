#!/bin/bash
# GPU box: VALU instruction counts + time per kernel for every tuning variant
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for f in $R/2d-weather-sandbox_amd/csrc/variants/libwxsim_*.so; do
  n=$(basename $f .so); rm -rf /tmp/pv
  WXSIM_LIB=$f rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d /tmp/pv -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1
  echo "== $n"; python $R/tools/rocpd_summary.py /tmp/pv/*.db --skip 2 | grep -E "fused|kernel \|" | sed 's/_ZN2wx9k_fused_\(.\)[^|]*/fused_\1 /' 
done
