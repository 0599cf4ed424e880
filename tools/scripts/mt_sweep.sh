# usage: bash tools/scripts/mt_sweep.sh [layer ...]   (on the GPU box through gpurun)
# Per-layer time with the block variant forced (EVFLY_WINO_MT = 1: 32-tile blocks, 2: 64-tile blocks) against the plan's own choice.
cd $GRAFT_REPO_ROOT
for mt in 0 1 2; do EVFLY_WINO_MT=$mt timeout 600 python tools/conv_sweep.py 200 "$@" 2>&1 | grep -v amdgpu | awk -v m=$mt '{printf "%s %s %s\n", m, $1, $2}'; done > gpurun_out/mt_sweep.txt
python3 - <<PY
import collections
d=collections.defaultdict(dict)
for l in open("gpurun_out/mt_sweep.txt"):
    p=l.split()
    if len(p)==3:
        try: d[p[1]][p[0]]=float(p[2])
        except: pass
print("layer   plan    MT=1    MT=2")
for k,v in d.items(): print(k, *["%.3f"%v.get(m,0) for m in "012"])
PY
