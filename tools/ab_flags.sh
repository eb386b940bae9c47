#!/bin/bash
# A/B of compile-time flags on the GPU box: tools/ab_flags.sh <tag> "<flags or empty>" ...   (each variant rebuilt on the box; CONFIG=cfg3 by default)
tag=$1; shift
cfg=${CONFIG:-cfg3}
i=0
for f in "$@"; do
  touch ppo_cpp_amd/csrc/ppo_hip.hip
  PPO_HIP_EXTRA_FLAGS="$f" python -m ppo_cpp_amd.build > /dev/null 2>&1
  python bench.py --config $cfg --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/abf_${tag}_$i.json 2> gpurun_out/abf_${tag}_$i.err
  python - "$f" gpurun_out/abf_${tag}_$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=d["kernels"]
    print("%-40s value %.4g step_us %.2f"%(sys.argv[1] or "(default)", d["value"], d["roofline"]["train_step"]["us_from_update_phase"]), {n:round(v["avg_us"],2) for n,v in k.items() if n in("train_fwd_bwd","weight_grad","grad_reduce","adam")})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
  i=$((i+1))
done
