#!/bin/bash
# diagnostic: rows-per-wave sweep of the default encode kernel
for r in 512 1024 2048 4096; do
  echo -n "rpi_max=$r  "
  PQHIP_DEBUG_RPI_MAX=$r python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'])"
done
