cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_models.py tests/test_gpu_bf16.py tests/test_gpu_standalone.py -x -q 2>&1 | tail -2
bash tools/scripts/layer_ab.sh 20 "EVFLY_IGEMM_NO_NMAP=1" "A=1" "EVFLY_IGEMM_NO_NMAP=1" "A=1" 2>&1 | sed 's/e12.*convlstm_x/... convlstm_x/' | cut -c1-200
for c in C2 C4 C3; do for e in "EVFLY_IGEMM_NO_NMAP=1" "A=1"; do echo "$c $e: $(env $e python3 bench.py --config $c --no-cpu-baseline --no-stage-rates --no-alt --no-other-configs 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['name']:x['ms_per_step'] for x in b['kernels']}
print(b['ms_per_step'], {n:k.get(n) for n in ('convlstm_h_gemm','convlstm_x_gemm','vit_linear','upconv2x2','vit_decoder_linear','lstm_x_gemm','vit_kv_reduce_conv')})")"; done; done
