# round 3: time the unit-kernel variants built by `tools/variants.sh build` (parity subset + hot-path bench each)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
bash tools/variants.sh run "fullsize_unit or inkernel_noise" 2>&1 | tee $O/variants.log
