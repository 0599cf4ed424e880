# usage: bash tools/scripts/gm_abl.sh     (on the GPU box through gpurun, after tools/scripts/build_variant.sh gmabl<bits> gconv.hip -DEVFLY_GM_ABL=<bits>)
# Timing of the MFMA grouped-conv kernel with parts compiled out (EVFLY_GM_ABL in gconv.hip; results of those builds are garbage).
cd $GRAFT_REPO_ROOT
for shape in "320 15 23 512" "320 8 12 1024" "320 15 23 128"; do for d in f32 bf16; do
  echo "== $shape $d"; python3 tools/probe_gconv.py $shape $d 200 | sed 's/rel_err.*//'
  for a in ${ABLS:-1 2 4 8 3}; do echo -n "abl $a: "; EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_gmabl$a.so python3 tools/probe_gconv.py $shape $d 200 | sed 's/(incl.*repack)//; s/rel_err.*//'; done
done; done
