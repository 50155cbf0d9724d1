# round-5 measurement suite (GPU box) -> gpurun_out/r05m/ (copied into profiles/ afterwards)
#   GPU test suite, default bench line, hot path at the four BASELINE shapes, rocprofv3 kernel stats of the hot path
#   and of the training step, PMC passes (VALU / LDS, FETCH / WRITE), per-step kernel breakdown
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
ulimit -c 0
if [ -z "$SKIP_TESTS" ]; then ( time python -m pytest tests -m gpu -x -q ) > $O/r05_gputest.log 2>&1; tail -3 $O/r05_gputest.log; fi
s=$(date +%s); python bench.py > $O/r05_bench_train_resnet18.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - s )) s" | tee $O/bench_default.time
for dm in smooth noise; do python bench.py --workload hotpath --disp $dm --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_hotpath_C2_$dm.json; done
python bench.py --workload hotpath --batch 4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_hotpath_C1.json
python bench.py --workload hotpath --batch 8 --height 320 --width 1024 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_hotpath_C4.json
python bench.py --workload hotpath --batch 12 --height 192 --width 512 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_hotpath_C5.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-leg > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-hotpath-leg --also-configs none --no-graph-leg --no-pmc-leg --no-mfma-leg --no-host-leg --no-kernel-leg > /dev/null 2>&1
cd $R
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r05_hotpath_kernel_stats.csv; rm -rf $O/hp
cp $(ls $O/tr/*/*kernel_stats.csv | head -1) $O/r05_train_kernel_stats.csv
python tools/step_breakdown.py $(ls $O/tr/*/*kernel_stats.csv | head -1) 9 > $O/r05_train_step_kernel_breakdown.csv 2>/dev/null; rm -rf $O/tr
head -5 $O/r05_hotpath_kernel_stats.csv | cut -c1-150
timeout 900 bash tools/pmc.sh > /dev/null; cp gpurun_out/pmc_summary.csv $O/r05_pmc_valu.csv
timeout 900 bash tools/pmc_traffic.sh > /dev/null; cp gpurun_out/pmc_traffic.csv $O/r05_pmc_fetch_write.csv
timeout 1200 bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.csv $O/r05_train_pmc_mfma.csv
grep -E "k_unit_fb" $O/r05_pmc_valu.csv $O/r05_pmc_fetch_write.csv | cut -c1-30,100-
head -4 $O/r05_train_step_kernel_breakdown.csv | cut -c1-120
# round-5 additions: fusion adjoint A/B (atomic scatter vs round-4 cell lists vs anchor lists) at realistic (6 px) and
# near-identity (0.3 px) flows, gradient-gap report against the float64 reference, unit-kernel skeleton / full variants
python tools/fusion_bwd_ab.py 6 > $O/r05_fusion_bwd_ab_6px.json 2>/dev/null
python tools/fusion_bwd_ab.py 0.3 > $O/r05_fusion_bwd_ab_03px.json 2>/dev/null
python tools/grad_error_report.py > $O/r05_grad_error_report.txt 2>/dev/null
if ls mono-vifi_amd/lib/var_*/libmvf_hotpath.so > /dev/null 2>&1; then VBENCH=--no-merge-unit-groups VUNITS=3 bash tools/variants.sh run gpurun_out/r05m/var fullsize_unit > /dev/null 2>&1; cp gpurun_out/r05m/var/variants.csv $O/r05_unit_kernel_variants.csv; fi
