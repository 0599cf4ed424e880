# developer sweep: two-stream pipeline == one-stream calls, bitwise, over chunk sizes / pipelines (tools/chunk_race_check.py)
cd $GRAFT_REPO_ROOT
run() { echo "== chunk $1 vit $2 dtype $3 B $4 T $5"; EVFLY_CHUNK_FRAMES=$1 RACE_VIT=$2 RACE_DTYPE=$3 python3 tools/chunk_race_check.py $4 $5 2>&1 | grep "pipelined\|serial" | cut -c1-140; }
run 80 base bf16 256 10
run 160 base bf16 256 10
run 320 base bf16 256 10
run 640 base bf16 256 10
run 640 tiny bf16 256 10
run 320 tiny bf16 64 16
run 640 tiny f32 64 5
run 160 tiny f32 64 5
run 640 base f32 256 5
run 320 base f32 256 5
