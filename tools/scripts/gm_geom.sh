# usage: bash tools/scripts/gm_geom.sh     (on the GPU box through gpurun): the MFMA grouped conv under forced tile geometries
cd $GRAFT_REPO_ROOT
for shape in "320 15 23 512" "320 8 12 1024" "320 15 23 128" "320 8 12 256"; do for d in f32 bf16; do
  echo "== $shape $d"; python3 tools/probe_gconv.py $shape $d 200 | sed 's/rel_err.*//'
  for rp in 1 2 4; do for fpb in 1 2 4; do echo -n "rparts $rp fpb $fpb: "; EVFLY_GM_RPARTS=$rp EVFLY_GM_FPB=$fpb python3 tools/probe_gconv.py $shape $d 200 2>&1 | tail -1 | sed 's/(incl.*repack)//; s/rel_err.*//'; done; done
done; done
