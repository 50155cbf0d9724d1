# round 3: time the unit-kernel variants built by `tools/variants.sh build` (parity subset + hot-path bench each),
# then their VALU instruction counts
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
bash tools/variants.sh run "${1:-fullsize_unit or inkernel_noise}" 2>&1 | tee $O/variants.log
rm -f $R/gpurun_out/r03f/pmc_variants.csv
bash tools/r03_pmc_variants.sh 2>&1 | grep -E "SQ_INSTS_VALU|GRBM_GUI" | tee -a $O/variants.log
