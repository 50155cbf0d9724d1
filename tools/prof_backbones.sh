# rocprofv3 per-step kernel breakdown of the DHRNet and Lite-Mono training steps (GPU box)
# -> gpurun_out/bk_<name>_{kernel_stats,step_breakdown}.csv
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() {   # name, bench flags...
  n=$1; shift
  timeout 900 bash tools/prof_train.sh "$@" > gpurun_out/bk_$n.log 2>&1
  cp gpurun_out/train_kernel_stats.csv gpurun_out/bk_${n}_kernel_stats.csv
  cp gpurun_out/train_step_kernel_breakdown.csv gpurun_out/bk_${n}_step_breakdown.csv
  tail -1 gpurun_out/bk_${n}_step_breakdown.csv; tail -1 gpurun_out/prof_train.log | cut -c1-160
}
mkdir -p gpurun_out
run dhrnet --backbone DHRNet
run litemono --backbone LiteMono --batch 8 --height 320 --width 1024
