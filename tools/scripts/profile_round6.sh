# usage: bash tools/scripts/profile_round6.sh <tag>      (on the GPU box through gpurun; writes gpurun_out/<tag>/)
# Round-6 evidence set, one gpurun call:
#   C2 (headline, fp32): default `python bench.py` line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
#   passes (roofline.traffic), SQ_VALU_MFMA_BUSY_CYCLES pass;  C5 and C3 (bf16 pipeline; C3 on one stream under the profiler): the same
#   four;  C3 / C4: bench lines.
# PMC passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section); the program goes
# straight behind `--`.
T=${1:-r6x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T; mkdir -p $O
# counter passes FIRST: their summaries are installed into profiles/ of this copy of the tree so that the bench lines below carry the
# `roofline.traffic` measured with the SAME code (bench.py reads profiles/r6_<config>_pmc_traffic.json)
for c in C2 C5 C3; do
  X=""; [ $c = C5 ] && X="--config C5"; [ $c = C3 ] && X="--config C3"
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $k -d $O/${c}_pmc_$k -o p --output-format csv -- python3 bench.py $X --steps 1 --warmup 1 --pmc-pass > $O/${c}_pmc_$k.line.json 2> $O/${c}_pmc_$k.err
  done
  # --pmc-pass: one stream, whole steps only; the step count is `steps_executed` of the line the profiled run printed, and the
  # summary refuses dispatch counts that are not a whole multiple of it
  python3 tools/pmc_traffic_summary.py $(ls $O/${c}_pmc_FETCH_SIZE/*counter_collection.csv | head -1) $(ls $O/${c}_pmc_WRITE_SIZE/*counter_collection.csv | head -1) $O/${c}_pmc_FETCH_SIZE.line.json $O/${c}_pmc_traffic.json && cp $O/${c}_pmc_traffic.json profiles/r6_${c}_pmc_traffic.json
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${c}_pmc_mfma -o p --output-format csv -- python3 bench.py $X --steps 2 --warmup 1 --pmc-pass > /dev/null 2> $O/${c}_pmc_mfma.err
  python3 tools/pmc_mfma_summary.py $(ls $O/${c}_pmc_mfma/*counter_collection.csv | head -1) $O/${c}_pmc_mfma.json
  rm -rf $O/${c}_pmc_FETCH_SIZE $O/${c}_pmc_WRITE_SIZE $O/${c}_pmc_mfma
done
# per-layer counters of the Winograd launches (C2, one stream): three SQ passes over whole steps, labelled by the bench line's launch order
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  BENCH_DUMP_LAYERS=1 rocprofv3 --kernel-trace --pmc $set -d $O/C2_layers_s$i -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --pmc-pass > $O/C2_layers_s$i.line.json 2> $O/C2_layers_s$i.err
done
python3 tools/pmc_wino_layers.py $O/C2_layers_s1.line.json $O/C2_wino_layers_pmc.json $(ls $O/C2_layers_s*/*counter_collection.csv) > $O/C2_wino_layers_pmc.txt 2>&1
cat $O/C2_wino_layers_pmc.txt
rm -rf $O/C2_layers_s1 $O/C2_layers_s2 $O/C2_layers_s3
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
