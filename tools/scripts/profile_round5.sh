# usage: bash tools/scripts/profile_round5.sh <tag>      (on the GPU box through gpurun; writes gpurun_out/<tag>/)
# Round-5 evidence set, one gpurun call:
#   C2 (headline, fp32): default `python bench.py` line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
#   passes (roofline.traffic), SQ_VALU_MFMA_BUSY_CYCLES pass;  C5 and C3 (bf16 pipeline; C3 on one stream under the profiler): the same
#   four;  C3 / C4: bench lines.
# PMC passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section); the program goes
# straight behind `--`.
T=${1:-r5x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T; mkdir -p $O
# counter passes FIRST: their summaries are installed into profiles/ of this copy of the tree so that the bench lines below carry the
# `roofline.traffic` measured with the SAME code (bench.py reads profiles/r5_<config>_pmc_traffic.json)
for c in C2 C5 C3; do
  X=""; [ $c = C5 ] && X="--config C5"; [ $c = C3 ] && X="--config C3"
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $k -d $O/${c}_pmc_$k -o p --output-format csv -- python3 bench.py $X --steps 1 --warmup 1 --pmc-pass > $O/${c}_pmc_$k.line.json 2> $O/${c}_pmc_$k.err
  done
  # --pmc-pass: one stream, whole steps only; the step count is `steps_executed` of the line the profiled run printed, and the
  # summary refuses dispatch counts that are not a whole multiple of it
  python3 tools/pmc_traffic_summary.py $(ls $O/${c}_pmc_FETCH_SIZE/*counter_collection.csv | head -1) $(ls $O/${c}_pmc_WRITE_SIZE/*counter_collection.csv | head -1) $O/${c}_pmc_FETCH_SIZE.line.json $O/${c}_pmc_traffic.json && cp $O/${c}_pmc_traffic.json profiles/r5_${c}_pmc_traffic.json
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${c}_pmc_mfma -o p --output-format csv -- python3 bench.py $X --steps 2 --warmup 1 --pmc-pass > /dev/null 2> $O/${c}_pmc_mfma.err
  python3 tools/pmc_mfma_summary.py $(ls $O/${c}_pmc_mfma/*counter_collection.csv | head -1) $O/${c}_pmc_mfma.json
  rm -rf $O/${c}_pmc_FETCH_SIZE $O/${c}_pmc_WRITE_SIZE $O/${c}_pmc_mfma
done
for c in C2 C5 C3; do
  X=""; [ $c = C5 ] && X="--config C5"; [ $c = C3 ] && X="--config C3 --no-overlap"
  # (the default C2 line carries the compact C5 / C3 / C4 objects under `other_configs`, exactly as the driver runs it)
  python3 bench.py $X > $O/${c}_bench.json 2> $O/${c}_bench.err
  rocprofv3 --kernel-trace --stats -d $O/${c}_prof -o p --output-format csv -- python3 bench.py $X --no-cpu-baseline --no-alt --no-stage-rates --no-other-configs > $O/${c}_bench_under_rocprof.json 2> $O/${c}_prof.err
  cp $(ls $O/${c}_prof/*kernel_stats.csv | head -1) $O/${c}_rocprofv3_kernel_stats.csv
  rm -rf $O/${c}_prof
  python3 tools/show_bench.py $O/${c}_bench.json | head -9
  head -6 $O/${c}_rocprofv3_kernel_stats.csv | cut -c1-200
done
for c in C3 C4; do
  python3 bench.py --config $c > $O/${c}_bench.json 2> $O/${c}_bench.err
  python3 tools/show_bench.py $O/${c}_bench.json | head -8
done
