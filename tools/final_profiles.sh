# Round-end artefacts: bench JSON lines and rocprofv3 kernel stats -> gpurun_out/final/
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final; mkdir -p $O
cd $R
python bench.py --workload hotpath --disp smooth --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_hotpath_smooth.json
python bench.py --workload hotpath --disp noise --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_hotpath_noise.json
python bench.py 2>/dev/null | tail -1 > $O/r01_bench_train.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r01_hotpath_kernel_stats.csv
cp $(ls $O/tr/*/*kernel_stats.csv | head -1) $O/r01_train_kernel_stats.csv
rm -rf $O/hp $O/tr
head -4 $O/r01_hotpath_kernel_stats.csv | cut -c1-150
cut -c1-400 $O/r01_bench_train.json
