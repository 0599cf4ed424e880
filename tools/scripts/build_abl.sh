# usage: bash tools/scripts/build_abl.sh <bits> [<bits> ...]     (in the build container, after `make -C evfly_amd/csrc`)
# Ablation builds of the Winograd kernel (compile-time EVFLY_WINO_ABL, see wino.hip): evfly_amd/libevfly_abl<bits>.so,
# selected at run time with EVFLY_LIB. Their results are garbage by construction; timing only (tools/scripts/abl_sweep.sh).
set -e
cd "$(dirname "$0")/../../evfly_amd/csrc"
OBJS="build/common.o build/cond.o build/igemm.o build/model.o build/ops.o build/remap.o build/voxel.o"
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DEVFLY_WINO_ABL=$a -c wino.hip -o build/wino_abl$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libevfly_abl$a.so $OBJS build/wino_abl$a.o
  echo "built evfly_amd/libevfly_abl$a.so"
done
