# usage: LIBS="prev hip" bash tools/scripts/lib_bench_ab.sh     (on the GPU box through gpurun)
# Whole-path A/B of library builds evfly_amd/libevfly_<name>.so: bench.py --steps 20 --warmup 5, three alternating repeats.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in ${LIBS:-prev hip}; do
  EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$l.so python bench.py --no-alt --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); k={x['name']:x['ms_per_step'] for x in b['kernels']}
print('$l', b['ms_per_step'], b['value'], 'lstm_h', k.get('convlstm_h_gemm'), 'gates', k.get('convlstm_gates'), 'lstm_x', k.get('convlstm_x_gemm'))"
done; done
