# parity tests + hot-path kernel times (smooth and noise disparity)
python -m pytest tests/test_hip_parity.py -q 2>&1 | tail -1
for d in smooth noise; do
python bench.py --workload hotpath --disp $d --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', {k:(v['avg_us'], v['frac']) for k,v in d['kernels'].items() if v}, d['value'])"
done
