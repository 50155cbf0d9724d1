# MFMA utilisation of the training step's kernels (north_star: "MFMA only for the conv GEMMs ...
# evidenced by rocprof ... MFMA utilisation").  Counters only (own run; no trace domains).
# usage (GPU box): bash tools/pmc_mfma.sh [extra bench.py flags] -> gpurun_out/pmc_mfma_summary.csv
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
# (no child legs inside a profiled run: a profiler around a run that starts profilers of its own hung the round-4 suite)
BENCH="timeout 600 python $R/bench.py --workload train --no-cpu-baseline --no-hotpath-leg --no-pmc-leg"
$BENCH --steps 2 --warmup 2 "$@" > /dev/null 2>&1      # warm the MIOpen find db of this box
rm -rf $R/gpurun_out/pmc_mfma
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- $BENCH --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R
python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma 3 > gpurun_out/pmc_mfma_summary.csv
rm -rf gpurun_out/pmc_mfma
head -32 gpurun_out/pmc_mfma_summary.csv | cut -c1-170
