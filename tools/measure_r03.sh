# round-3 measurement suite (GPU box): GPU test suite, the default bench line, rocprofv3 kernel stats of the
# hot path, PMC passes (VALU / LDS, FETCH / WRITE) -> gpurun_out/r03g/ (copied into profiles/ afterwards)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
ulimit -c 0
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1; tail -4 $O/gputest.log
s=$(date +%s); python bench.py > $O/r03_bench_train_resnet18.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - s )) s" | tee $O/bench_default.time
for dm in smooth noise; do python bench.py --workload hotpath --disp $dm --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_hotpath_C2_$dm.json; done
python bench.py --workload hotpath --batch 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_hotpath_C1.json
python bench.py --workload hotpath --batch 8 --height 320 --width 1024 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_hotpath_C4.json
python bench.py --workload hotpath --batch 12 --height 192 --width 512 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_hotpath_C5.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r03_hotpath_kernel_stats.csv; rm -rf $O/hp
head -5 $O/r03_hotpath_kernel_stats.csv | cut -c1-150
bash tools/pmc.sh > /dev/null; cp gpurun_out/pmc_summary.csv $O/r03_pmc_valu.csv
bash tools/pmc_traffic.sh > /dev/null; cp gpurun_out/pmc_traffic.csv $O/r03_pmc_fetch_write.csv
grep -E "k_unit_fb" $O/r03_pmc_valu.csv $O/r03_pmc_fetch_write.csv | cut -c1-30,100-
