cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size" 2>&1 | tail -15
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('measured_copy_GBps'), d['roofline'].get('valu'))"
