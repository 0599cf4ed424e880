# effective shader clock of k_wino8 on layer e32 with and without its memory operations (gpurun box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in 0 29; do
  EVFLY_WINO_ABL=$a timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/clk_$a -o p --output-format csv -- python3 tools/conv_probe.py e32 3 > /dev/null 2>&1
done
