#!/bin/bash
# sample GPU clock / power (rocm-smi) while a bench workload runs; usage: tools/clk_sample.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
grep -o "\"value\": [0-9.e+]*\|\"ms_per_step\": [0-9.]*" /tmp/clk_bench.json
