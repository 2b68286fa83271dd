#!/bin/bash
# GPU box: the round-3 evidence in one call -> gpurun_out/r3p/ (copied into profiles/ by hand). Sections can be selected: tools/r03_profiles.sh "probe ablation ..."
# (tuning environment switches exist only in the -DWX_DEBUG build of the library: make -C 2d-weather-sandbox_amd/csrc debug)
export WXSIM_LIB=${WXSIM_LIB:-$GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc/variants/libwxsim_debug.so}
[ -f "$WXSIM_LIB" ] || make -C $GRAFT_REPO_ROOT/2d-weather-sandbox_amd/csrc debug
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R; export TMPDIR=/tmp
WHAT=${1:-"probe lottery ubench sweep slabs flow trace_wet trace_dry"}
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has probe; then
  { echo "tools/alloc_probe.py: handles of the same 16384x2048 grid alive at once, same binary, each timed alone (3 interleaved repetitions of 100 iterations):";
    echo "avg kernel ms of march_wet_full_iteration, device addresses of five planes. arena 0 = one hipMalloc per plane (rounds 1-2), arena 1 = one allocation per handle;";
    echo "third field 1 = hipDeviceMallocContiguous (physically contiguous VRAM)";
    python tools/alloc_probe.py 0:0 0:0 0:0 0:0 1:0 1:0 1:0 1:0 1:0:1 1:0:1 1:4096:1 1:99618816:1 2>&1 | grep -v amdgpu.ids;
    echo; echo "wx_tune_placement on a fresh handle (WX_TUNE_DEBUG=1), then the sustained kernel time:";
    WX_TUNE_DEBUG=1 WX_PROBE_TUNE=1 python tools/alloc_probe.py 1:0 2>&1 | grep -v amdgpu.ids;
    WX_TUNE_DEBUG=1 WX_PROBE_TUNE=1 python tools/alloc_probe.py 1:0 2>&1 | grep -v amdgpu.ids; } > $O/alloc_probe.txt 2>&1
fi
if has lottery; then python tools/copy_lottery.py > $O/copy_lottery.txt 2>&1; fi
if has ubench; then { for i in 1 2 3; do tools/ubench_hbm 16384 2048 64 15; done; tools/ubench_hbm 32768 4096 32 15; tools/ubench_hbm 16384 2048 256 15; } > $O/ubench_hbm.txt 2>&1; fi
if has sweep; then { python tools/shape_sweep_whole.py 16384 2048 "4x1,1x0.5,1x0.25" "3x1,1x0.5,1x0.25" "2x1,1x0.5,1x0.25" "5x1,1x0.5,1x0.25"; python tools/shape_sweep_whole.py 32768 4096 "7x1,1x0.5,1x0.25,1x0.125" "6x1,1x0.5,1x0.25" "5x1,1x0.5,1x0.25"; python tools/slab_sweep.py; } 2>&1 | grep -v amdgpu.ids > $O/shape_sweep.txt; fi
if has slabs; then { python tools/slab_shapes.py 48 3; python tools/slab_shapes.py 42 3; python tools/slab_shapes.py 24 3; } 2>&1 | grep -v amdgpu.ids > $O/slab_shapes.txt; fi
if has flow; then sed -i 's/for sigma in ([^)]*):/for sigma in (0.05,0.1,0.15,0.2,0.3,0.4):/' tools/flow_probe.py; python tools/flow_probe.py 2>&1 | grep -v amdgpu.ids > $O/flow_probe.txt; fi
if has trace_wet; then bash tools/prof_bench.sh r3p/wet_march > $O/wet_march_console.txt 2>&1; fi
if has trace_dry; then BENCH_ARGS="--workload dry --X 32768 --Y 4096" bash tools/prof_bench.sh r3p/dry_march > $O/dry_march_console.txt 2>&1; fi
if has ablation; then BENCH_ARGS="--flow 0 --tune 0" STEPS=40 bash tools/wet_variants.sh > $O/wet_ablation.txt 2>&1; fi
if has particles; then bash tools/prof_particles.sh r3p/particles > $O/particles_console.txt 2>&1; fi
# keep the summaries, drop the raw rocprofv3 databases (gpurun copies back at most 64 MiB)
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls $O
