#!/usr/bin/env python3
"""Where one env step of the host-Env leg goes on the GPU side: parses a rocprofv3 --kernel-trace --memory-copy-trace run of
`tools/hostenv_timeline.py run` (PPO2::learn, 4096 SeededEnvMock behind VecEnv, [256,256]) and prints, per env step of the steady state, the
durations of the H2D copy, norm_batch_kernel, policy_step_kernel and the D2H copy and the gaps between them.
usage (on the GPU box):  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 tools/hostenv_timeline.py run
                         python3 tools/hostenv_timeline.py parse gpurun_out/tl"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    from ppo_cpp_amd import hostapi
    r = hostapi.learn(int(os.environ.get("E", "4096")), 16, [256, 256], n_updates=4, nminibatches=32, noptepochs=1)
    print(r["collect_ms"], r["phase_ms"])
    sys.exit(0)
d = sys.argv[2]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "norm_batch" if "norm_batch_kernel" in n else "policy_step" if "policy_step_kernel" in n else None
        if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dirn = r.get("Direction", "")
        k = "h2d" if "HOST_TO_DEVICE" in dirn else "d2h" if "DEVICE_TO_HOST" in dirn else None
        if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
# an env step of the steady state: h2d -> norm_batch -> policy_step -> d2h, back to back
seqs = []
for i in range(len(ev) - 3):
    if [e[2] for e in ev[i:i + 4]] == ["h2d", "norm_batch", "policy_step", "d2h"]: seqs.append(ev[i:i + 4])
seqs = seqs[len(seqs) // 2:]                                  # the later updates
if not seqs: sys.exit("no h2d -> norm_batch -> policy_step -> d2h sequence found (%d events)" % len(ev))
import statistics as st
us = lambda a: st.median(a) / 1e3
print("%d env steps" % len(seqs))
print("  H2D copy            %6.1f us" % us([s[0][1] - s[0][0] for s in seqs]))
print("  gap                 %6.1f us" % us([s[1][0] - s[0][1] for s in seqs]))
print("  norm_batch_kernel   %6.1f us" % us([s[1][1] - s[1][0] for s in seqs]))
print("  gap                 %6.1f us" % us([s[2][0] - s[1][1] for s in seqs]))
print("  policy_step_kernel  %6.1f us" % us([s[2][1] - s[2][0] for s in seqs]))
print("  gap                 %6.1f us" % us([s[3][0] - s[2][1] for s in seqs]))
print("  D2H copy            %6.1f us" % us([s[3][1] - s[3][0] for s in seqs]))
print("  H2D start -> D2H end %5.1f us" % us([s[3][1] - s[0][0] for s in seqs]))
if len(seqs) > 1: print("  D2H end -> next H2D start (host: copy out, Env::step, pack) %6.1f us" % us([b[0][0] - a[3][1] for a, b in zip(seqs, seqs[1:]) if b[0][0] - a[3][1] < 1e6]))
