cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in ${VARIANTS:-fused fullmaps nofuse}; do
    unset EVFLY_NO_SKIP_FUSION EVFLY_FULL_ENCODER_OUTPUTS; if [ $v = nofuse ]; then export EVFLY_NO_SKIP_FUSION=1; fi; if [ $v = fullmaps ]; then export EVFLY_FULL_ENCODER_OUTPUTS=1; fi
    python bench.py --no-alt --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print('$v', b['ms_per_step'], b['value'])"
  done
done
