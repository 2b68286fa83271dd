#!/bin/bash
# GPU box: rocprofv3 evidence of round 6 with the SHIPPED library (no tuning switches) -> gpurun_out/r6p/ (summaries copied into profiles/r06_*).
# Sections: trace_wet trace_dry particles driver   (tools/r06_profiles.sh "trace_wet driver")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O; cd $R; export TMPDIR=/tmp
WHAT=${1:-"trace_wet trace_dry particles driver"}
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has driver; then python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; fi
if has trace_wet; then bash tools/prof_bench.sh r6p/wet_march > $O/wet_march_console.txt 2>&1; fi
if has trace_dry; then BENCH_ARGS="--workload dry --X 32768 --Y 4096" bash tools/prof_bench.sh r6p/dry_march > $O/dry_march_console.txt 2>&1; fi
if has particles; then bash tools/prof_particles.sh r6p/particles > $O/particles_console.txt 2>&1; fi
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls $O
