#!/bin/bash
# A/B of a build switch at cfg3: tools/r3_ab.sh <tag> [ENV=1 ...]   (bench line per variant under gpurun_out/)
tag=$1; shift
mkdir -p gpurun_out
for v in base "$@"; do
  if [ "$v" = base ]; then python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/ab_${tag}_base.json 2> gpurun_out/ab_${tag}_base.err
  else env $v python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/ab_${tag}_${v%%=*}.json 2> gpurun_out/ab_${tag}_${v%%=*}.err; fi
done
python - <<'PY' $tag
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/ab_%s_*.json"%sys.argv[1])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels"]
        print(f.split("/")[-1], "value %.4g"%d["value"], "upd_ms %.3f"%d["phase_ms"]["update"], "step_us %.2f"%d["roofline"]["train_step"]["us_from_update_phase"], {n:round(v["avg_us"],2) for n,v in k.items() if n in("train_fwd_bwd","weight_grad","grad_reduce","adam")})
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
