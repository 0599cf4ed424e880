# usage (GPU box): bash tools/scripts/coop_ab.sh     -- the cooperative ConvLSTM (k_clstm16_coop) against the paths it replaces
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "convlstm" 2>&1 | tail -5
if [ -f evfly_amd/libevfly_cots.so ]; then
EVFLY_LIB=evfly_amd/libevfly_cots.so timeout 300 python tools/clstm_ts.py 20 16 2>&1 | grep -v amdgpu.ids | tail -10
EVFLY_LIB=evfly_amd/libevfly_cots.so timeout 300 python tools/clstm_ts.py 64 10 2>&1 | grep -v amdgpu.ids | tail -10
fi
for i in 1 2; do
for c in C5 C3; do
echo "== $c coop"; timeout 300 python bench.py --config $c --no-cpu --no-stage-rates 2>gpurun_out/err_$c.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], [ (k['name'],k['ms_per_step']) for k in d['kernels'] if 'lstm' in k['name']])" || tail -5 gpurun_out/err_$c.txt
echo "== $c old"; EVFLY_NO_CLSTM16_COOP=1 timeout 300 python bench.py --config $c --no-cpu --no-stage-rates 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], [ (k['name'],k['ms_per_step']) for k in d['kernels'] if 'lstm' in k['name']])"
done; done
