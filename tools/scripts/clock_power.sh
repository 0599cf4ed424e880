# usage: bash tools/scripts/clock_power.sh <layer> <reps>   -- sclk / power (rocm-smi) sampled while the Winograd kernel loops on one layer
cd $GRAFT_REPO_ROOT
L=${1:-e32}; R=${2:-1500}
python3 tools/conv_sweep.py $R $L &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr '\n' ';'; echo
  sleep 0.4
done
wait $PID
