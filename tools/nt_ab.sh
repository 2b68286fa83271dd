L=$GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc/variants/libwxsim_debug.so
run() { WXSIM_LIB=$L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['value']), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
for nt in 0 1; do echo "== wet 16384x2048 WX_NT=$nt"; WX_NT=$nt run python $GRAFT_REPO_ROOT/bench.py --steps 200 --no-north-star --no-extras --no-cpu-baseline --no-pmc; done
for nt in 0 1; do echo "== dry 32768x4096 WX_NT=$nt"; WX_NT=$nt run python $GRAFT_REPO_ROOT/bench.py --workload dry --X 32768 --Y 4096 --steps 200 --no-north-star --no-extras --no-cpu-baseline --no-pmc; done
for nt in 0 1; do echo "== dry 4096x1024 WX_NT=$nt"; WX_NT=$nt run python $GRAFT_REPO_ROOT/bench.py --workload dry --X 4096 --Y 1024 --steps 1000 --warmup 200 --no-north-star --no-extras --no-cpu-baseline --no-pmc; done
for nt in 0 1; do echo "== wet 4096x1024 WX_NT=$nt"; WX_NT=$nt run python $GRAFT_REPO_ROOT/bench.py --X 4096 --Y 1024 --steps 1000 --warmup 200 --no-north-star --no-extras --no-cpu-baseline --no-pmc; done
for nt in 0 1; do echo "== wet 8192x2048 WX_NT=$nt"; WX_NT=$nt run python $GRAFT_REPO_ROOT/bench.py --X 8192 --Y 2048 --steps 400 --warmup 100 --no-north-star --no-extras --no-cpu-baseline --no-pmc; done
done
