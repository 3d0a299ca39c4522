#!/bin/bash
# lookup (select + reconstruct + rescale): split code matrix + scale vector vs interleaved records of 20 / 32 bytes,
# for a resident matrix that fits the 256 MB Infinity Cache (10 M rows) and one that does not (100 M rows)
for NC in 10000000 100000000; do
  for mode in "" "--record-bytes 20" "--record-bytes 32"; do
    for i in 1 2; do
      python bench.py --workload lookup $mode --lookup-codes $NC --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('codes=$NC $mode:', '%.4g vec/s' % r['value'], 'frac %.3f' % r['roofline']['frac'], 'min %.3f max %.3f ms' % (r['roofline']['min_launch_ms'], r['roofline']['max_launch_ms']))"
    done
  done
done
