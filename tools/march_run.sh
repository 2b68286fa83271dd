cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for i in 1 2; do
for c in 0 1; do
WX_COW=$c timeout 300 python bench.py --no-cpu-baseline --no-pmc --steps 200 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cow=$c', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
done
