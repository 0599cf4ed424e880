# usage: bash tools/scripts/pmc_kernel.sh <outdir> <kernel-name pattern> <program> [args...]   (on the GPU box through gpurun)
# SQ / LDS / cache counter passes (one rocprofv3 --pmc run per set, kernel-trace only) over any probe command; per-kernel means
# of the dispatches whose name contains the pattern (tools/pmc_summary.py). The program goes straight behind `--`.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$1; P=$2; shift 2
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/s$i -o p --output-format csv -- "$@" > /dev/null 2> $O.err$i
  f=$(ls $O/s$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f "$P"
done
