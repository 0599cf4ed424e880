# usage: bash tools/scripts/r3_benches.sh <tag> [configs...]     (on the GPU box through gpurun; writes gpurun_out/<tag>/)
# One driver-form bench line per BASELINE config: C2 (headline, default flags), C5, C3, C4-shard.
T=${1:-r3x}; shift
CFGS=${@:-C2 C5 C3 C4}
O=gpurun_out/$T; mkdir -p $O
for c in $CFGS; do
  if [ $c = C2 ]; then python3 bench.py > $O/bench_C2.json 2> $O/bench_C2.err
  else python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; fi
  echo "== $c rc=$?"; tail -c 300 $O/bench_$c.err | grep -v amdgpu.ids
  python3 tools/show_bench.py $O/bench_$c.json 2>&1 | head -${SHOW:-12}
done
