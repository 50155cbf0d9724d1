# round 6, second GPU batch: parity of the pruned unit kernel with the 32-wide adjoint walk, its LDS conflict share,
# and the fast-mode report (exact vs fast at C1 / C2 / C4 / C5)
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -4 > $O/t_parity.log
python bench.py --workload hotpath --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/hp_C2_exact.json
python tools/fast_mode_report.py > $O/r06_fast_mode_report.json 2> $O/fast.err
cat $O/t_parity.log; cat $O/hp_C2_exact.json | cut -c1-900; tail -3 $O/fast.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06b/r06_fast_mode_report.json'))
for k,v in d['shapes'].items(): print(k, v.get('exact'), v.get('fast'), v.get('fast_over_exact_time'), v.get('deviation'))
PY
