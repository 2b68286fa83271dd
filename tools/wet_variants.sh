#!/bin/bash
# GPU box: bench time + SQ counters of the dominant kernel for every tuning variant under csrc/variants/ (and the default build).
# usage: tools/wet_variants.sh [kernel-name regex, default march_wet] ; env BENCH_ARGS, WX_FUSED (default 2)
R=$GRAFT_REPO_ROOT; PAT=${1:-march_wet}; cd /tmp; export TMPDIR=/tmp; export WX_FUSED=${WX_FUSED:-2}
for f in $R/2d-weather-sandbox_amd/csrc/libwxsim.so $R/2d-weather-sandbox_amd/csrc/variants/libwxsim_*.so; do
  [ -f $f ] || continue
  n=$(basename $f .so); echo "== $n"
  WXSIM_LIB=$f python $R/bench.py --steps ${STEPS:-60} --warmup 6 --no-cpu-baseline --no-pmc --no-north-star --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms', {a:round(b,4) for a,b in k.items()})"
  [ -n "$NOPMC" ] && continue
  rm -rf /tmp/pv1 /tmp/pv2
  WXSIM_LIB=$f rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pv1 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras $BENCH_ARGS > /dev/null 2>&1
  WXSIM_LIB=$f rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d /tmp/pv2 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras $BENCH_ARGS > /dev/null 2>&1
  if [ -n "$TRAFFIC" ]; then
    for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pv3
      WXSIM_LIB=$f rocprofv3 --pmc $c -d /tmp/pv3 -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras $BENCH_ARGS > /dev/null 2>&1
      python $R/tools/rocpd_summary.py /tmp/pv3/*.db --skip 2 | grep -E "$PAT" | sed 's/_ZN2wx[0-9]*\(k_[a-z_]*\)[^|]*/\1 /' | tail -2 | sed "s/^/    $c KiB: /"
    done
  fi
  for d in /tmp/pv1 /tmp/pv2; do python $R/tools/rocpd_summary.py $d/*.db --skip 2 | grep -E "$PAT|kernel \| SQ" | sed 's/_ZN2wx[0-9]*\(k_[a-z_]*\)[^|]*/\1 /' | tail -2; done
done
