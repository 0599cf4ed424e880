# usage: bash tools/scripts/build_w4.sh     (in the build container, after `make -C evfly_amd/csrc`)
# Developer library evfly_amd/libevfly_w4.so = the product objects + the Winograd F(4x4,3x3) prototype (tools/proto/wino4.hip), with
# evfly_op_conv2d_nhwc routed to it under EVFLY_WINO4=1 (model.hip built with -DEVFLY_WITH_WINO4). Used by tools/scripts/w4_ab.sh and
# tests/test_gpu_wino.py::test_wino4_prototype (skipped when the library is absent). The product library never contains this kernel.
set -e
cd "$(dirname "$0")/../../evfly_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -I../../include"
/opt/rocm/bin/hipcc $F -c ../../tools/proto/wino4.hip -o build/wino4_proto.o
/opt/rocm/bin/hipcc $F -DEVFLY_WITH_WINO4 -c model.hip -o build/model_w4.o
OBJS=$(for f in *.hip; do b=${f%.hip}; [ "$b" = model ] || echo build/$b.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libevfly_w4.so $OBJS build/model_w4.o build/wino4_proto.o
echo "built evfly_amd/libevfly_w4.so"
