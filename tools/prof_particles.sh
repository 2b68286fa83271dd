#!/bin/bash
# GPU-box script: kernel-trace stats + HBM traffic passes for the particle workload (BASELINE configs[4] on one GPU) after a
# warm-up long enough for the droplet pool to be active. Outputs under gpurun_out/<tag>_*.
R=$GRAFT_REPO_ROOT; TAG=${1:-particles}; O=$R/gpurun_out; N=${PARTICLES:-1048576}
ARGS="--particles $N --no-cpu-baseline --no-pmc --no-north-star --no-extras"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o trace -- python $R/bench.py $ARGS --steps 100 --warmup 500 > $O/${TAG}_trace_bench.json 2> $O/${TAG}.err
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o p -- python $R/bench.py $ARGS --tune 0 --steps 10 --warmup 500 > /dev/null 2>> $O/${TAG}.err
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o p -- python $R/bench.py $ARGS --tune 0 --steps 10 --warmup 500 > /dev/null 2>> $O/${TAG}.err
python $R/tools/rocpd_summary.py $O/${TAG}_trace/*.db --last 100 > $O/${TAG}_trace.md 2>&1
for d in pmc_fetch pmc_write; do python $R/tools/rocpd_summary.py $O/${TAG}_$d/*.db --skip 500 > $O/${TAG}_$d.md 2>&1; done
cat $O/${TAG}_trace_bench.json | cut -c1-400; tail -n +1 $O/${TAG}_trace.md | head -14; for d in pmc_fetch pmc_write; do tail -9 $O/${TAG}_$d.md; done; grep -i "error\|fail" $O/${TAG}.err | head -5
