# usage: bash tools/scripts/build_variant.sh <name> <file.hip> "<-D flags>"     (in the build container, after `make -C evfly_amd/csrc`)
# A/B build of one translation unit with extra defines: evfly_amd/libevfly_<name>.so, selected at run time with EVFLY_LIB
# (tools/scripts/lib_bench_ab.sh).
set -e
cd "$(dirname "$0")/../../evfly_amd/csrc"
name=$1; src=$2; defs=$3; base=${src%.hip}
extra=""; [ "$base" = wino ] && extra="-fno-slp-vectorize"
OBJS=$(for f in *.hip; do b=${f%.hip}; [ "$b" = "$base" ] || echo build/$b.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra $defs -c $src -o build/${base}_var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libevfly_$name.so $OBJS build/${base}_var_$name.o
echo "built evfly_amd/libevfly_$name.so"
