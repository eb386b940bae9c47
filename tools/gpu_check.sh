#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_check.sh <tag>   -> tests + bench + rocprof kernel stats under gpurun_out/<tag>/
tag=${1:-run}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $out/tests.log; grep -E "passed|failed|Error" $out/tests.log | tail -3
python bench.py --steps 5 --warmup 2 ${BENCH_ARGS} > $out/bench.json 2> $out/bench.err || tail -5 $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print("value %.4g env-steps/s  ms/step %.2f  phases %s" % (d["value"], d["ms_per_step"], d["phase_ms"]))
print("roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k!="train_step"}, d["roofline"]["train_step"])
for k,v in d["kernels"].items(): print("  %-16s %8.2f us x %d" % (k, v["avg_us"], v["launches"]))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_prof.json 2> $out/prof.err
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; cut -d, -f1-4 $f | head -8
