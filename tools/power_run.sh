#!/bin/bash
# run a command in the background and sample sclk / socket power beside it
"$@" > /tmp/power_run.out 2>&1 &
BP=$!
sleep 1.5
for i in $(seq 1 6); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
tail -2 /tmp/power_run.out
