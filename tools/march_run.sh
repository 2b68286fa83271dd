cd /root/repo
V=/root/repo/2d-weather-sandbox_amd/csrc/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "marching" 2>&1 | tail -2
for i in 1 2; do
for lib in "" $V/libwxsim_noxcd.so; do
WXSIM_LIB=$lib timeout 300 python bench.py --workload dry --no-cpu-baseline --no-pmc --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done
done
for lib in "" $V/libwxsim_noxcd.so; do
WXSIM_LIB=$lib timeout 300 python bench.py --workload dry --X 32768 --Y 4096 --no-cpu-baseline --no-pmc --steps 50 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp
for lib in "" $V/libwxsim_noxcd.so; do
rm -rf /tmp/pv; WXSIM_LIB=$lib rocprofv3 --pmc FETCH_SIZE -d /tmp/pv -o p -- python /root/repo/bench.py --workload dry --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py /tmp/pv/*.db --skip 2 | grep -E "march" | tail -1 | sed 's/_ZN2wx11k_march[^|]*/march/'
done
