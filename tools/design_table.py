#!/usr/bin/env python3
"""Fills the round-4 table of DESIGN.md (between the R4_TABLE markers) from profiles/pmc_traffic.json, so that the
numbers in the document are the committed records, not a transcription.  usage: python tools/design_table.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
PEAK_TF, PEAK_GBS = 157.3, 8000.0
names = [("encode@10000000@d300_m15_k256", "PQ encode 10 M × 300, M = 15 (configs[1])", "mfma", 2 * 256 * 300),
         ("opq_encode@10000000@d300_m15_k256", "OPQ rotate + encode 10 M × 300 (configs[2])", "mfma", 2 * 300 * 300 + 2 * 256 * 300),
         ("reconstruct@100000000@d300_m15_k256", "PQ reconstruct 100 M codes (configs[3])", "hbm", 1215),
         ("encode_d768@12500000@d768_m48_k256", "PQ encode 12.5 M × 768, M = 48 (configs[4] shard)", "mfma", 2 * 256 * 768),
         ("reconstruct@10000000@d300_m15_k256", "PQ reconstruct 10 M codes", "hbm", 1215),
         ("opq_reconstruct@10000000@d300_m15_k256", "OPQ reconstruct 10 M codes (gather inside the rotation)", "mfma", 2 * 300 * 300),
         ("lookup@10000000@d300_m15_k256", "lookup, 10 M of 10 M resident rows", "hbm", 1227),
         ("lookup@10000000@d300_m15_k256_codes100000000", "lookup, 10 M of 100 M resident rows (two passes)", "hbm", 1227),
         ("adc_scan@100000000@d300_m15_k256", "ADC scan 100 M rows, 1 query", "hbm", 19),
         ("adc_scan@100000000@d300_m15_k256_q8", "ADC scan 100 M rows, 8 queries", "hbm", 47),
         ("encode@10000000@d128_m16_k16", "PQ encode d = 128, M = 16, K = 16 (benches/pq.rs shape)", "hbm", 528),
         ("encode@10000000@d20_m10_k128", "PQ encode d = 20, M = 10, K = 128 (pq.rs:431-440 test shape; candidate lists)", "hbm", 90),
         ("encode@10000000@d300_m150_k256", "PQ encode d = 300, M = 150, K = 256 (two-float sub-vectors; candidate lists)", "hbm", 1350),
         ("reconstruct@10000000@d300_m150_k256", "PQ reconstruct 10 M codes, d = 300, M = 150 (grouped LDS gather)", "hbm", 1350),
         ("kmeans@10000000@d300_m15_k256", "k-means iteration, 15 subquantizers, 10 M × 300", "mfma", 2 * 256 * 300),
         ("opq_train@10000000@d300_m15_k256", "OPQ training step, exact cross product", "mfma", 2 * 300 * 300 * 2 + 2 * 2 * 256 * 300),
         ("opq_train@10000000@d300_m15_k256_fastcross", "OPQ training step, float-tolerance cross product", "mfma", 2 * 300 * 300 * 2 + 2 * 2 * 256 * 300)]
rows = ["| workload | dominant kernel | rocprof ms (timed launches) | events ms / step | fraction of roofline | HBM traffic ÷ algorithmic | in-kernel clock |", "|---|---|---|---|---|---|---|"]
for key, label, bound, per in names:
    e = E.get(key)
    if not e:
        continue
    k = re.sub(r"^void pqhip::", "", e["kernel"]).split("(")[0]
    ms = e["bench_ms_per_step_same_run"]
    n = e["rows"]
    frac = (per * n / (ms * 1e-3) / 1e12 / PEAK_TF) if bound == "mfma" else (per * n / (ms * 1e-3) / 1e9 / PEAK_GBS)
    clk = e.get("in_kernel_clock_mhz_diag_pass")
    tl = e.get("avg_launch_ms_timed_launches")
    rows.append("| %s | `%s` | %s | %.3f | **%.3f** of %s | %s | %s |" % (
        label, k[:60], ("%.3f" % tl) if tl else "—", ms, frac, "FP32 MFMA" if bound == "mfma" else "HBM",
        ("%.2f" % e["ratio"]) if e.get("ratio") else "—", ("%.0f MHz" % clk) if clk else "—"))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
i, j = s.index("<!-- R4_TABLE_BEGIN -->"), s.index("<!-- R4_TABLE_END -->")
s = s[:i] + "<!-- R4_TABLE_BEGIN -->\n" + "\n".join(rows) + "\n" + s[j:]
open(p, "w").write(s)
print("\n".join(rows))
