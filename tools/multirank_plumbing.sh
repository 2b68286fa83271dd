#!/bin/bash
# GPU box (one GPU): the N-rank bench path end to end with every rank on cuda:0 -- gloo transport (host-staged halos), then an attempt
# with RCCL itself (it normally refuses two ranks on one device; if it does not, this is first contact with the nccl path).
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
run() { echo "== $*"; env WX_BENCH_SHARE_GPU=1 "$@" > /tmp/plumb.out 2>&1; rc=$?; grep -a '^{"metric"' /tmp/plumb.out | tail -1 | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ', {k: d.get(k) for k in ('value','n_gpus','ranks_seen','verify','transport','ms_per_step','scaling')}, d.get('config',{}).get('workload','')[:80])"; [ $rc -ne 0 ] && { echo "rc=$rc"; grep -i "error\|Traceback\|raise\|WxError\|assert" /tmp/plumb.out | head -12 | cut -c1-300; }; sleep 5; }
# (round 5) the command as the driver may type it: NO launcher -- bench.py re-executes itself under torch.distributed.run; --verify is on by default
run WX_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 40 --warmup 8
# (four ranks of a 32768x4096 grid share ONE GPU here: --tune 2 keeps the placement search from filling its memory with candidate sets)
run WX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 8
run WX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 40 --warmup 8 --X 32768 --Y 4096 --tune 2
run WX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 36 --warmup 9 --particles 262144
run WX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 4 --steps 40 --warmup 8 --workload dry --X 32768 --Y 4096 --verify --tune 2
run WX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29516 bench.py --gpus 2 --steps 40 --warmup 8 --verify
run WX_DIST_BACKEND=nccl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 40 --warmup 8
