#!/bin/bash
# GPU-box script: kernel-trace stats + PMC passes for the default bench workload. Outputs under gpurun_out/<tag>_*.
R=$GRAFT_REPO_ROOT; TAG=${1:-fused}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (the trace run tunes the placement like the plain bench run does: the kernel's duration depends on it; the counters do not)
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o trace -- python $R/bench.py $BENCH_ARGS --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-north-star --no-extras --no-arith-fast > $O/${TAG}_trace_bench.json 2> $O/${TAG}.err
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o p -- python $R/bench.py $BENCH_ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras --no-arith-fast --tune 0 > /dev/null 2>> $O/${TAG}.err
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o p -- python $R/bench.py $BENCH_ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras --no-arith-fast --tune 0 > /dev/null 2>> $O/${TAG}.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/${TAG}_pmc_sq1 -o p -- python $R/bench.py $BENCH_ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras --no-arith-fast --tune 0 > /dev/null 2>> $O/${TAG}.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d $O/${TAG}_pmc_sq2 -o p -- python $R/bench.py $BENCH_ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-extras --no-arith-fast --tune 0 > /dev/null 2>> $O/${TAG}.err
python $R/tools/rocpd_summary.py $O/${TAG}_trace/*.db --last 100 > $O/${TAG}_trace.md 2>&1
for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2; do python $R/tools/rocpd_summary.py $O/${TAG}_$d/*.db --skip 2 > $O/${TAG}_$d.md 2>&1; done
tail -n +3 $O/${TAG}_trace.md | head -8; for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2; do tail -6 $O/${TAG}_$d.md; done; grep -i "error\|fail" $O/${TAG}.err | head -5
