R=$GRAFT_REPO_ROOT
for f in $R/2d-weather-sandbox_amd/csrc/variants/libwxsim_*.so; do
 for ms in 32 48 64 96; do
  n=$(basename $f .so)
  for sz in "32768 4096 100" "16384 2048 300"; do set -- $sz
  WX_MARCH_MAXSEG=$ms WXSIM_LIB=$f python $R/bench.py --workload dry --X $1 --Y $2 --steps $3 --warmup 10 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n maxseg=$ms $1x$2', round(d['value']), round(d['roofline']['frac'],4))"
  done
 done
done
