#!/bin/bash
# A/B of prebuilt library variants at configs[4] (cfg5): tools/ab_variants_cfg5.sh <tag> <lib or "default"> ...   three alternating rounds
tag=$1; shift
mkdir -p gpurun_out/abv
for r in 1 2 3; do
  for v in "$@"; do
    name=$(basename $v .so)
    if [ "$v" = "default" ]; then unset PPO_HIP_LIBRARY; else export PPO_HIP_LIBRARY=$PWD/$v; fi
    python bench.py --config cfg5 --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/abv/${tag}_${name}_$r.json 2> gpurun_out/abv/${tag}_${name}_$r.err
    python - "$name" $r gpurun_out/abv/${tag}_${name}_$r.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]); k=d["kernels"]
    print("%-14s run %s value %.4g ms_per_step %.2f step_us %.2f"%(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("train_step",{}).get("us_from_update_phase",-1)), {n:round(v["avg_us"],2) for n,v in k.items() if n in("train_fwd_bwd","weight_grad","grad_reduce","adam")})
except Exception as e: print(sys.argv[1:], "ERR", e)
PY
  done
done
