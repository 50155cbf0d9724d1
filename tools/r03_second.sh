# round 3, second GPU pass: (1) MIOpen user find-db for the four BASELINE training configurations, written
# under gpurun_out/ so it travels back (to be committed under mono-vifi_amd/miopen_db/); cold / warm
# start times of the bench; (2) rocprofv3 kernel stats + PMC passes of the hot path.
# usage (GPU box): bash tools/r03_second.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
ulimit -c 0
export MIOPEN_USER_DB_PATH=$R/gpurun_out/miopen_db; mkdir -p $MIOPEN_USER_DB_PATH
run() { n=$1; shift; s=$(date +%s); python bench.py --no-cpu-baseline --no-hotpath-leg --also-configs none --steps 10 --warmup 5 "$@" 2> $O/$n.err | tail -1 > $O/$n.json; echo "$n $(( $(date +%s) - s )) s" >> $O/times.log; }
run cold_C2
run cold_C3 --backbone DHRNet
run cold_C4 --backbone LiteMono --batch 8 --height 320 --width 1024
run cold_C5 --backbone DHRNet --height 192 --width 512
ls -la $MIOPEN_USER_DB_PATH >> $O/times.log; du -sh ~/.cache/miopen ~/.config/miopen 2>/dev/null >> $O/times.log
run warm_C2
# a fresh box that has the find-db but not the compiled-kernel cache
rm -rf ~/.cache/miopen
run dbonly_C2
run dbonly_C3 --backbone DHRNet
du -sh ~/.cache/miopen 2>/dev/null >> $O/times.log
cat $O/times.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r03_hotpath_kernel_stats.csv; rm -rf $O/hp
head -6 $O/r03_hotpath_kernel_stats.csv | cut -c1-200
bash tools/pmc.sh; cp gpurun_out/pmc_summary.csv $O/r03_pmc_valu.csv
bash tools/pmc_traffic.sh; cp gpurun_out/pmc_traffic.csv $O/r03_pmc_fetch_write.csv
