#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name up to the argument list), per counter, mean per launch.
usage: pmc_summary.py out.json note pass1.csv [pass2.csv ...]"""
import collections, csv, json, sys
out, note, files = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {"note": note, "kernels": {}}
for k, d in agg.items():
    e = {c: round(v / cnt[(k, c)], 1) for c, v in d.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
        wc = 4.0 * e["SQ_WAVE_CYCLES"]                      # SQ_WAVE_CYCLES counts quad-cycles
        e["derived"] = {"mfma_busy_cycles_over_wave_cycles": round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / wc, 4)}
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e: e["derived"][c.lower() + "_frac"] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = round((2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res["kernels"]), "kernels")
