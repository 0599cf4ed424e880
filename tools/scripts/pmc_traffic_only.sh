cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p1; mkdir -p $O
for c in C2 C5 C3; do
  X=""; [ $c = C5 ] && X="--config C5"; [ $c = C3 ] && X="--config C3"
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $k -d $O/${c}_pmc_$k -o p --output-format csv -- python3 bench.py $X --steps 1 --warmup 1 --pmc-pass > $O/${c}_pmc_$k.line.json 2> $O/${c}_pmc_$k.err
  done
  python3 tools/pmc_traffic_summary.py $(ls $O/${c}_pmc_FETCH_SIZE/*counter_collection.csv | head -1) $(ls $O/${c}_pmc_WRITE_SIZE/*counter_collection.csv | head -1) $O/${c}_pmc_FETCH_SIZE.line.json $O/${c}_pmc_traffic.json
  python3 - <<PY
import csv, collections
for k in ("FETCH_SIZE","WRITE_SIZE"):
    import glob
    f=glob.glob("$O/${c}_pmc_%s/*counter_collection.csv"%k)[0]
    c=collections.Counter(r["Kernel_Name"][:90] for r in csv.DictReader(open(f)))
    print(k, {n:v for n,v in c.items() if v%3})
PY
  rm -rf $O/${c}_pmc_FETCH_SIZE $O/${c}_pmc_WRITE_SIZE
done
