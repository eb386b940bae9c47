#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>"   (one --pmc pass, kernel-trace only; summarised per kernel name)
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $out/pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_bench.json 2> $out/pmc.err || tail -5 $out/pmc.err
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items(): print("    %-32s %16.1f per-launch" % (c, v / cnt[(k, c)]))
PY
