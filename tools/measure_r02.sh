# Round-2 measurement suite (GPU box): bench lines per BASELINE.md section 5 + rocprofv3 summaries.
# Writes gpurun_out/r02/*; copy what should be judged into profiles/.
#   bash tools/measure_r02.sh [hotpath|train|prof|all]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02; mkdir -p $O
what=${1:-all}
cd $R
if [ $what = hotpath ] || [ $what = all ]; then
  # hot path only: the four BASELINE shapes (smooth disparity) + C2 with i.i.d.-noise disparity
  python bench.py --workload hotpath --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/r02_bench_hotpath_C2_smooth.json
  python bench.py --workload hotpath --steps 50 --warmup 10 --disp noise --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_hotpath_C2_noise.json
  python bench.py --workload hotpath --steps 50 --warmup 10 --noise tensor --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_hotpath_C2_noise_tensor.json
  python bench.py --workload hotpath --steps 50 --warmup 10 --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_hotpath_C1.json
  python bench.py --workload hotpath --steps 50 --warmup 10 --batch 8 --height 320 --width 1024 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_hotpath_C4.json
  python bench.py --workload hotpath --steps 50 --warmup 10 --batch 12 --height 192 --width 512 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_hotpath_C5.json
  for f in $O/r02_bench_hotpath_*.json; do python -c "import json,sys; d=json.load(open('$f')); k=d['kernels']['unit_fwdbwd']; print('$(basename $f)', d['value'], 'img/s', k['avg_us'], 'us', k['achieved'], 'GB/s frac', k['frac'])"; done
fi
if [ $what = train ] || [ $what = all ]; then
  python bench.py 2>/dev/null | tail -1 > $O/r02_bench_train_resnet18.json
  python bench.py --backbone DHRNet --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_train_dhrnet_640x192.json
  python bench.py --backbone LiteMono --batch 8 --height 320 --width 1024 --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_train_litemono_1024x320.json
  python bench.py --backbone DHRNet --height 192 --width 512 --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_train_dhrnet_cs_512x192.json
  for f in $O/r02_bench_train_*.json; do python -c "import json; d=json.load(open('$f')); print('$(basename $f)', d['value'], 'img/s', d['ms_per_step'], 'ms')"; done
fi
if [ $what = prof ] || [ $what = all ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  cd $R
  cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r02_hotpath_kernel_stats.csv; rm -rf $O/hp
  head -6 $O/r02_hotpath_kernel_stats.csv | cut -c1-160
  bash tools/pmc.sh > /dev/null; cp gpurun_out/pmc_summary.csv $O/r02_pmc_valu.csv
  bash tools/pmc_traffic.sh > /dev/null; cp gpurun_out/pmc_traffic.csv $O/r02_pmc_fetch_write.csv
  bash tools/prof_train.sh > /dev/null 2>&1
  cp gpurun_out/train_kernel_stats.csv $O/r02_train_kernel_stats.csv
  cp gpurun_out/train_step_kernel_breakdown.csv $O/r02_train_step_kernel_breakdown.csv
  bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.csv $O/r02_train_pmc_mfma.csv
  tail -1 $O/r02_train_step_kernel_breakdown.csv; tail -1 $O/r02_train_pmc_mfma.csv
fi
