# round 6, third GPU batch: same-box A/B of the unit kernel (round-5 source vs this round's), the per-configuration gradient
# error counts, the eight-rank rehearsal and the remaining GPU tests
O=gpurun_out/r06c; mkdir -p $O; rm -f gpurun_out/grad_error_counts.tsv
VUNITS=4.5 bash tools/variants.sh run $O/ab > $O/ab.log 2>&1
VUNITS=4.5 bash tools/variants.sh run $O/ab2 > $O/ab2.log 2>&1
python -m pytest tests/test_hip_parity.py -q -m gpu -k "fullsize" 2>&1 | tail -3 > $O/t_fullsize.log
cp gpurun_out/grad_error_counts.tsv $O/ 2>/dev/null
python -m pytest tests/test_bench_launch.py -q -m gpu 2>&1 | tail -5 > $O/t_launch.log
cat $O/ab/variants.csv $O/ab2/variants.csv | cut -c1-200; cat $O/t_fullsize.log $O/t_launch.log; cat $O/grad_error_counts.tsv
