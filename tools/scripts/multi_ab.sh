# usage: LIBS="a b" [SWEEP=tools/gemm_sweep.py] bash tools/scripts/multi_ab.sh      (on the GPU box through gpurun)
# Two alternating repeats of a sweep per library evfly_amd/libevfly_<name>.so (the first run on a box reads ~1 % slow).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in ${LIBS:-r2d hip}; do EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$l.so timeout 600 python ${SWEEP:-tools/conv_sweep.py} 200 2>&1 | grep -v amdgpu | awk -v l=$l '{printf "%s %s %s\n", l, $1, $2}' ; done; done
