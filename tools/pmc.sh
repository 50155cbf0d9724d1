# PMC passes (counters only, own runs; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: bash tools/pmc.sh   (on the GPU box; writes gpurun_out/pmc_*/ and gpurun_out/pmc_summary.csv)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
HP="python $R/bench.py --workload hotpath --steps 4 --warmup 2 --no-cpu-baseline --no-pmc-leg"
TR="python $R/bench.py --workload train --steps 2 --warmup 2 --no-cpu-baseline --no-pmc-leg"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_hp1 -- $HP > $R/gpurun_out/pmc_hp1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_hp2 -- $HP > $R/gpurun_out/pmc_hp2.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_hp1 gpurun_out/pmc_hp2 | grep -E "pass,|k_photo|k_unit_fb|k_units_|k_fb_" > gpurun_out/pmc_summary.csv
rm -rf gpurun_out/pmc_hp1 gpurun_out/pmc_hp2   # raw traces are large; the summary travels back
grep -c . gpurun_out/pmc_summary.csv
