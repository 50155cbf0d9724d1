R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r02h; mkdir -p $O
ulimit -c 0
( timeout 900 python -m pytest tests/ -q -m gpu ) > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py 2> $O/bench_train.err | tail -1 > $O/r02_bench_train_resnet18.json
python -c "import json; d=json.load(open('$O/r02_bench_train_resnet18.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_us'])"
timeout 600 bash tools/prof_train.sh > /dev/null 2>&1
cp gpurun_out/train_kernel_stats.csv $O/r02_train_kernel_stats.csv
cp gpurun_out/train_step_kernel_breakdown.csv $O/r02_train_step_kernel_breakdown.csv
grep -i "pool\|TOTAL" $O/r02_train_step_kernel_breakdown.csv | cut -c1-140
rm -rf gpurun_out/prof_train
