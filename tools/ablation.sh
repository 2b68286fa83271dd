R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for f in $R/2d-weather-sandbox_amd/csrc/variants/libwxsim_*.so; do
  n=$(basename $f .so); rm -rf /tmp/pv
  WXSIM_LIB=$f rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pv -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1
  echo "== $n"; python $R/tools/rocpd_summary.py /tmp/pv/*.db --skip 2 | grep -E "fused" | sed 's/_ZN2wx9k_fused_\(.\)[^|]*/fused_\1 /' | tail -2
  WXSIM_LIB=$f python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
