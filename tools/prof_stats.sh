#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py at one config: tools/prof_stats.sh <tag> <config> [ENV=1 ...]  -> gpurun_out/prof_<tag>_stats.csv
tag=$1; cfg=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $out/prof_${tag}_bench.json 2> $out/prof_${tag}.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $out/prof_${tag}_stats.csv
# kernels launched at several sizes (the staging kernel per env step and per epoch): median duration per (kernel, grid size)
python3 - "$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in ("stage", "gather", "norm_batch")): d[(n, r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, g), v in sorted(d.items()): print("  by size: %-40s grid %8s calls %5d median_us %8.2f" % (n[:40], g, len(v), statistics.median(v) / 1e3))
PY
python3 - "$out/prof_${tag}_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg_us %8.2f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
