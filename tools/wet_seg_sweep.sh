R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('   ', round(d['value']), 'Mcs/s', round(d['ms_per_step'],4), 'ms', {a:round(b,4) for a,b in k.items()})"; }
for a in 1.0 2.0; do for sgr in 256 228 205 128 114 76 57 50 44; do echo "seg_rows=$sgr alpha=$a"; WX_WET_SEG=$sgr WX_WET_ALPHA=$a run; done; done
