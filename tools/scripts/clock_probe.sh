# effective shader clock of the Winograd kernel on layer e32 with and without its memory operations (gpurun box):
# GRBM_GUI_ACTIVE over the kernel's duration. Needs evfly_amd/libevfly_abl29.so (tools/scripts/build_abl.sh 29).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in hip abl29; do
  EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$lib.so timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/clk_$lib -o p --output-format csv -- python3 tools/conv_probe.py e32 3 > /dev/null 2>&1
done
