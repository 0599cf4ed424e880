# usage (GPU box): bash tools/scripts/pool_skip_ab.sh   -- k16_pool_bilinear (levels 3 / 4 of the bf16 U-Net) against the pool + resize launches, strip heights
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for c in C5 C3; do
for v in 2 4 8 12 old; do
E="EVFLY_POOL_SKIP_ROWS=$v"; [ $v = old ] && E="EVFLY_NO_POOL_SKIP_FUSION=1"
echo -n "== $c $v: "; env $E timeout 300 python bench.py --config $c --no-cpu --no-stage-rates 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], [ (k['name'],k['ms_per_step']) for k in d['kernels'] if k['name'] in ('maxpool','skip_bilinear')])"
done; done; done
