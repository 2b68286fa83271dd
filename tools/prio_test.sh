R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms', {a:round(b,4) for a,b in k.items()})"; }
export -f run; export R
for cfg in "WX_WET_ROUNDS=4" "WX_WET_ROUNDS=1 WX_WET_ALPHA=1.0 WX_WET_SKEW=0" "WX_WET_ROUNDS=1 WX_WET_ALPHA=1.0 WX_WET_SKEW=0.2" "WX_WET_ROUNDS=1 WX_WET_ALPHA=1.0 WX_WET_SKEW=0.3" "WX_WET_ROUNDS=1 WX_WET_ALPHA=1.0 WX_WET_SKEW=0.4" "WX_WET_ROUNDS=1 WX_WET_ALPHA=1.2 WX_WET_SKEW=0.3" "WX_WET_ROUNDS=2 WX_WET_ALPHA=1.0 WX_WET_SKEW=0.3" "WX_WET_ROUNDS=4 WX_WET_ALPHA=1.0 WX_WET_SKEW=0.3" "WX_WET_ROUNDS=4 WX_WET_ALPHA=2.0 WX_WET_SKEW=0.3"; do
  echo " $cfg"; env $cfg bash -c run
done
echo TIMING; WX_WET_ROUNDS=1 WX_WET_ALPHA=1.0 WX_WET_SKEW=0.3 WXSIM_LIB=$R/2d-weather-sandbox_amd/csrc/variants/libwxsim_timing.so python $R/bench.py --steps 45 --warmup 0 --frame 45 --no-cpu-baseline --no-pmc --no-north-star 2>&1 >/dev/null | grep -E "seg "
