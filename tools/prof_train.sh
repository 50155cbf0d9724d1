# rocprofv3 kernel stats of the training step -> gpurun_out/prof_train/ + per-step breakdown
# usage: bash tools/prof_train.sh [extra bench.py flags]
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train
python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-hotpath-leg "$@" > /dev/null 2>&1   # warm MIOpen find db (kept in ~/.config/miopen on this box)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-hotpath-leg "$@" > $R/gpurun_out/prof_train.log 2>&1
cd $R
rm -f gpurun_out/prof_train/*/*kernel_trace.csv
f=$(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/train_kernel_stats.csv
python tools/step_breakdown.py $f 9 70 > gpurun_out/train_step_kernel_breakdown.csv
head -60 gpurun_out/train_step_kernel_breakdown.csv | cut -c1-150
tail -1 gpurun_out/train_step_kernel_breakdown.csv
tail -1 gpurun_out/prof_train.log | cut -c1-200
