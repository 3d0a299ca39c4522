#!/bin/bash
# every bench workload once (round-3 evidence): default line (headline + configs[2..4]) and the other workloads
OUT=gpurun_out/$1; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for w in reconstruct opq_reconstruct lookup kmeans opq_train; do
  python bench.py --workload $w --steps 5 --warmup 1 > $OUT/$w.json 2> $OUT/$w.err
done
python bench.py --workload adc_scan --steps 10 > $OUT/adc_scan_q1.json 2> $OUT/adc_scan_q1.err
python bench.py --workload adc_scan --queries 8 --steps 10 > $OUT/adc_scan_q8.json 2> $OUT/adc_scan_q8.err
python bench.py --workload adc_scan --k 1024 --rows 20000000 --steps 5 > $OUT/adc_scan_k1024.json 2> $OUT/adc_scan_k1024.err
python bench.py --workload encode --d 128 --m 16 --k 16 --steps 10 --no-sub-configs > $OUT/smallk.json 2> $OUT/smallk.err
python bench.py --workload encode --d 20 --m 10 --k 128 --steps 10 --no-sub-configs > $OUT/refshape_d20.json 2> $OUT/refshape_d20.err
python bench.py --in-process 1 --rows 4000000 --steps 3 > $OUT/inproc1.json 2> $OUT/inproc1.err
python bench.py --in-process 2 --single-device --rows 4000000 --steps 3 > $OUT/inproc2.json 2> $OUT/inproc2.err
python - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e); continue
    roof = r.get("roofline") or {}
    cb = r.get("cpu_baseline") or {}
    par = {k: v for k, v in cb.items() if k.startswith("gpu_")}
    print(os.path.basename(f), "%.4g" % r["value"], r["unit"], "frac", roof.get("frac") and round(roof["frac"], 3), "ms", roof.get("avg_launch_ms") and round(roof["avg_launch_ms"], 3), par, r.get("summary", "") and json.dumps(r["summary"]))
PY
