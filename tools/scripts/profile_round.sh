# usage: bash tools/scripts/profile_round.sh <tag>      (on the GPU box through gpurun; writes gpurun_out/<tag>_*)
# bench line, rocprofv3 kernel stats of the same command, and the FETCH_SIZE / WRITE_SIZE passes bench.py's
# roofline.traffic reads (separate --pmc runs, kernel-trace only: see MI355X_MICROARCH.md, HBM section).
T=${1:-rX}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-alt > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.err
cp $(ls $O/${T}_prof/*kernel_stats.csv | head -1) $O/${T}_rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/${T}_pmc_$c -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > /dev/null 2> $O/${T}_pmc_$c.err
done
# bench.py runs warmup + 1 untimed profiled step + steps timed steps = 3 steps in all here
python3 tools/pmc_traffic_summary.py $(ls $O/${T}_pmc_FETCH_SIZE/*counter_collection.csv | head -1) $(ls $O/${T}_pmc_WRITE_SIZE/*counter_collection.csv | head -1) 3 $O/${T}_pmc_traffic.json
python3 tools/show_bench.py $O/${T}_bench.json | head -50
head -8 $O/${T}_rocprofv3_kernel_stats.csv | cut -c1-160
