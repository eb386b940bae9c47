#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_cfg.sh <tag> <config> [extra bench args]  -> bench line + rocprofv3 kernel stats of one config
tag=$1; cfg=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py --config $cfg --steps 3 --warmup 1 --no-extra "$@" > $out/bench_$cfg.json 2> $out/bench_$cfg.err || tail -5 $out/bench_$cfg.err
python - <<PY
import json
d=json.load(open("$out/bench_$cfg.json"))
print("$cfg value %.4g env-steps/s  ms/step %.2f  phases %s dtype %s" % (d["value"], d["ms_per_step"], d["phase_ms"], d["dtype"]))
print("roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k not in ("train_step","timing")}, d["roofline"]["train_step"])
for k,v in d["kernels"].items(): print("  %-16s %8.2f us x %d" % (k, v["avg_us"], v["launches"]))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("vectorised",{}).get("value"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/bench_prof_$cfg.json 2> $out/prof_$cfg.err
f=$(find $out/prof_$cfg -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$cfg.csv; cut -d, -f1-4 $f | head -14
