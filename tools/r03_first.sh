# round 3, first GPU pass: new batched-unit tests, the parity suite, hot-path kernel timings of the
# launch-structure variants.   usage (GPU box): bash tools/r03_first.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
python -m pytest tests/test_units_batched.py -x -q 2>&1 | tail -15 > $O/units_batched.log
python -m pytest tests/test_hip_parity.py -q -x 2>&1 | tail -15 > $O/parity.log
for v in "" "--no-share-identity" "--no-batch-units"; do
  for dm in smooth noise; do
    python bench.py --workload hotpath --disp $dm --steps 30 --warmup 5 --no-cpu-baseline $v 2> $O/err.log | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']['unit_fwdbwd']; print('variant[$v] $dm', 'launch_us', k['avg_us'], 'us_per_unit', k['us_per_unit'], 'frac', k['frac'], 'img/s', d['value'], 'graph', d.get('hip_graph_replay'))" >> $O/hotpath.log 2>&1
  done
done
cat $O/units_batched.log $O/parity.log $O/hotpath.log
