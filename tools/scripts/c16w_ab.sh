# usage (GPU box, through gpurun): bash tools/scripts/c16w_ab.sh [reps] [layer ...]
# Same-box A/B of the bf16 deep-layer kernels on the U-Net layer shapes at 320 frames (tools/probe16.py): the wide-tile kernel
# (conv16w.hip) against the 128 x 128 implicit GEMM (EVFLY_NO_CONV16W=1), optionally forcing the channel-tile width.
cd $GRAFT_REPO_ROOT
REPS=${1:-200}; shift
LAYERS=${@:-e32 e41 e42 e51 e52 d11 d12 d21 d22 d31}
for l in $LAYERS; do
  a=$(python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1)
  b=$(EVFLY_NO_CONV16W=1 python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1)
  c=$(EVFLY_CONV16W_BC=128 python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1)
  e=$(EVFLY_CONV16W_BC=256 python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1)
  echo "new: $a | old: $b | bc128: $c | bc256: $e"
done
