# usage (GPU box): bash tools/scripts/attn_ab.sh   -- attention in the query projection's epilogue (igemm16 OUT_ATTN) against the two launches
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -x -q -m gpu -k "attention or vit or c3" 2>&1 | tail -4
for i in 1 2; do
for v in new old; do
E="X=1"; [ $v = old ] && E="EVFLY_NO_ATTN_FUSION=1"
echo -n "== C3 $v: "; env $E timeout 300 python bench.py --config C3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['stage_rates']['p_only']['ms'], [ (k['name'],k['ms_per_step']) for k in d['kernels'] if k['name'] in ('vit_linear','vit_attention')])"
done; done
