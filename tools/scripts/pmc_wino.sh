# usage: bash tools/scripts/pmc_wino.sh <layer> <outdir> [kernel-name pattern]   (run on the GPU box through gpurun)
# Counter passes over the Winograd kernel on one layer (tools/conv_probe.py); EVFLY_LIB selects the build.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=${1:-e32}; O=${2:-gpurun_out/pmc_wino}; P=${3:-k_wino}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/s$i -o p --output-format csv -- python3 tools/conv_probe.py $L 3 > /dev/null 2>&1
  f=$(ls $O/s$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f $P
done
