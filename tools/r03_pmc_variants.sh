# round 3: VALU instruction counts of the unit-kernel variants (rocprofv3 --pmc, counters only)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in $R/mono-vifi_amd/lib/var_*; do
  n=$(basename $d); export MVF_HOTPATH_LIB=$d/libmvf_hotpath.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --workload hotpath --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  ( cd $R && python tools/pmc_summary.py gpurun_out/r03f/pmc_$n | grep -E "k_unit_fb" | sed "s/^/$n,/" ) >> $O/pmc_variants.csv
  rm -rf $O/pmc_$n
done
cat $O/pmc_variants.csv | cut -c1-40,110-
