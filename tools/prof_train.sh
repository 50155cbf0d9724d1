# rocprofv3 kernel stats of the training step -> gpurun_out/prof_train/ (kernel trace removed: large)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/prof_train/*/*kernel_trace.csv
f=$(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1); head -${1:-45} $f | cut -c1-160
tail -1 gpurun_out/prof_train.log | cut -c1-200
