# Counters of the fusion adjoint's anchor-list gather (k_fusion_level_bwd_anchor), level 0 of the ResNet18 pyramid at the
# merged batch of a step (B 36, 64 channels, 96 x 320): one counter-only rocprofv3 pass per group around tools/anchor_probe.py.
#   bash tools/anchor_pmc.sh [flow sigma px] -> gpurun_out/anchor_pmc.csv  
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A=${1:-6}
O=$R/gpurun_out/anchor_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/anchor_probe.py $A > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed" >> $O/failed.txt
done
python $R/tools/pmc_summary.py $O/p* | grep -i "pass,\|anchor" > $R/gpurun_out/anchor_pmc.csv
python $R/tools/anchor_probe.py $A 2>/dev/null | tail -1 >> $R/gpurun_out/anchor_pmc.csv
cat $R/gpurun_out/anchor_pmc.csv; cat $O/failed.txt 2>/dev/null
