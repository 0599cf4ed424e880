# developer bisect: which kernel family makes the pipelined depth differ at 320-frame chunks (tools/chunk_race_check.py)
cd $GRAFT_REPO_ROOT
export EVFLY_CHUNK_FRAMES=${CH:-320}
for sw in NONE EVFLY_NO_MIXFFN16 EVFLY_NO_CLSTM16_SEQ EVFLY_NO_CONV16W EVFLY_NO_CONV16 EVFLY_NO_CONV16W_UP EVFLY_NO_CONV16W_GEMM EVFLY_NO_OUT16_FUSION EVFLY_NO_E11_FUSION EVFLY_NO_SKIP_FUSION EVFLY_VOX_NO_OPTIMISTIC; do
  echo "== $sw"
  env $sw=1 python3 tools/chunk_race_check.py 2>&1 | grep "pipelined step [12]" | cut -c1-150
done
