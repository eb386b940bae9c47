#!/usr/bin/env python3
"""Per-launch durations of ONE train step from a rocprofv3 --kernel-trace CSV: finds the last span between two launches that end a train step (adam_kernel, or bf16_reduce_adam_kernel on the bf16 path) and prints every launch
in it in order (kernels that appear several times per step -- the bf16 path's GEMMs -- are told apart by their position).  Median over the last N steps.
usage: python tools/step_sequence.py <kernel_trace.csv> [steps]"""
import csv, statistics, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ends = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel") or r[2].startswith("bf16_reduce_adam_kernel")]
steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])]
steps = [s for s in steps if s and len(s) == len(steps[-1]) and [k[2] for k in s] == [k[2] for k in steps[-1]]][-n_steps:]
print("%d identical steps of %d launches" % (len(steps), len(steps[-1])))
tot = 0.0
for j, k in enumerate(steps[-1]):
    d = statistics.median(s[j][1] - s[j][0] for s in steps) / 1e3
    gap = statistics.median((s[j][0] - s[j - 1][1]) for s in steps) / 1e3 if j else 0.0
    tot += d
    print("  %2d %-44s %8.2f us   (gap before %5.2f)" % (j, k[2][:44], d, gap))
print("  sum of kernels %.1f us ; step (first start -> last end) %.1f us" % (tot, statistics.median(s[-1][1] - s[0][0] for s in steps) / 1e3))
