# Round-6 measurement set (GPU box): bash tools/measure_r06.sh -> gpurun_out/r06/  (copied to profiles/r06_* by hand)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
rm -f gpurun_out/grad_error_counts.tsv
# 1. the driver's default command, twice (cpu_baseline stability), then with the detail legs
for k in 1 2; do s=$(date +%s); python bench.py > $O/bench_default_$k.out 2> $O/bench_default_$k.err; echo "default $k rc=$? $(( $(date +%s) - s )) s" >> $O/times.log; done
s=$(date +%s); python bench.py --detail --detail-out $O/r06_bench_detail.json > $O/bench_detail.out 2> $O/bench_detail.err; echo "detail rc=$? $(( $(date +%s) - s )) s" >> $O/times.log
tail -1 $O/bench_default_1.out > $O/r06_bench_train_resnet18.json
# 2. the hot path alone at the four BASELINE shapes (+ noise disparity)
for dm in smooth noise; do python bench.py --workload hotpath --disp $dm --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_hotpath_C2_$dm.json; done
python bench.py --workload hotpath --batch 4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_hotpath_C1.json
python bench.py --workload hotpath --batch 8 --height 320 --width 1024 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_hotpath_C4.json
python bench.py --workload hotpath --batch 12 --height 192 --width 512 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_hotpath_C5.json
# 3. rocprofv3 kernel stats of the same commands (hot path, training step)
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline --no-pmc-leg > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-hotpath-leg --no-pmc-leg > /dev/null 2>&1 )
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r06_hotpath_kernel_stats.csv
cp $(ls $O/tr/*/*kernel_stats.csv | head -1) $O/r06_train_kernel_stats.csv
python tools/step_breakdown.py $O/r06_train_kernel_stats.csv 9 > $O/r06_train_step_kernel_breakdown.csv 2>/dev/null
rm -rf $O/hp $O/tr
# 4. counters of the unit kernel (separate passes), MFMA / VALU of the step's kernels
bash tools/pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.csv $O/r06_pmc_valu.csv
bash tools/pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic.csv $O/r06_pmc_fetch_write.csv
bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.csv $O/r06_train_pmc_mfma.csv 2>/dev/null
# 5. fast-mode report
python tools/fast_mode_report.py > $O/r06_fast_mode_report.json 2> $O/fast.err
# 6. the GPU suite + smoke
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/r06_gputest.log
cp gpurun_out/grad_error_counts.tsv $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gputest.log 2>&1
cat $O/times.log; tail -3 $O/r06_gputest.log; wc -c $O/bench_default_1.out; tail -1 $O/bench_default_1.out | cut -c1-1500
