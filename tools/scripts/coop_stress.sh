# usage (GPU box): bash tools/scripts/coop_stress.sh   -- eight two-stream C3 runs (k_clstm16_coop beside the ViT-base velocity model's kernels) + the GPU suite
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --config C3 --no-cpu --no-stage-rates --steps 10 --warmup 2 2>gpurun_out/c3_err_$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" || { echo "FAIL $i"; tail -8 gpurun_out/c3_err_$i.txt; }; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
