# HBM traffic of the tile kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots),
# per MI355X_MICROARCH.md.  Writes gpurun_out/pmc_traffic.csv (mean KB per launch).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
HP="python $R/bench.py --workload hotpath --steps 4 --warmup 2 --no-cpu-baseline --no-pmc-leg"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -- $HP > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- $HP > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w | grep -E "pass,|k_photo|k_unit_fb|k_units_|k_fb_|k_disp_mean" > gpurun_out/pmc_traffic.csv
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
cat gpurun_out/pmc_traffic.csv
