# Round-end measurement suite (GPU box): full GPU test suite, bench lines of the four training configurations,
# rocprofv3 kernel stats of the hot path and of the ResNet18 / DHRNet steps -> gpurun_out/r02f/ (copy into profiles/).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=gpurun_out/r02f; mkdir -p $O
ulimit -c 0
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gputest.log 2>&1; tail -4 $O/gputest.log
timeout 900 python bench.py 2> $O/bench_train.err | tail -1 > $O/r02_bench_train_resnet18.json
timeout 900 python bench.py --backbone DHRNet --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_train_dhrnet_640x192.json
timeout 900 python bench.py --backbone LiteMono --batch 8 --height 320 --width 1024 --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_train_litemono_1024x320.json
timeout 900 python bench.py --backbone DHRNet --height 192 --width 512 --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_train_dhrnet_cs_512x192.json
for f in $O/r02_bench_train_*.json; do python -c "import json; d=json.load(open('$f')); print('$(basename $f)', d['value'], 'img/s', d['ms_per_step'], 'ms', (d.get('roofline') or {}).get('avg_us'))"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r02_hotpath_kernel_stats.csv; rm -rf $O/hp
head -4 $O/r02_hotpath_kernel_stats.csv | cut -c1-160
timeout 600 bash tools/prof_train.sh > /dev/null 2>&1
cp gpurun_out/train_kernel_stats.csv $O/r02_train_kernel_stats.csv
cp gpurun_out/train_step_kernel_breakdown.csv $O/r02_train_step_kernel_breakdown.csv
tail -1 $O/r02_train_step_kernel_breakdown.csv
timeout 600 bash tools/prof_train.sh --backbone DHRNet > /dev/null 2>&1
cp gpurun_out/train_step_kernel_breakdown.csv $O/r02_train_step_kernel_breakdown_dhrnet.csv
tail -1 $O/r02_train_step_kernel_breakdown_dhrnet.csv
rm -rf gpurun_out/prof_train
