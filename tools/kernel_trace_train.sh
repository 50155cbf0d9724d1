R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/kt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-hotpath-leg --no-pmc-leg > /dev/null 2>&1
cd $R
cp $(ls $O/tr/*/*kernel_stats.csv | head -1) $O/r04_train_kernel_stats.csv
python tools/step_breakdown.py $(ls $O/tr/*/*kernel_stats.csv | head -1) 9 > $O/r04_train_step_kernel_breakdown.csv 2>/dev/null; rm -rf $O/tr
head -6 $O/r04_train_step_kernel_breakdown.csv | cut -c1-120
