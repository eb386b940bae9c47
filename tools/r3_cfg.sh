#!/bin/bash
# bench one config with optional env switches: tools/r3_cfg.sh <config> [ENV=1 ...] -> one line per variant
cfg=$1; shift
for v in base "$@"; do
  if [ "$v" = base ]; then python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/cfg_${cfg}_base.json 2> gpurun_out/cfg_${cfg}_base.err; f=gpurun_out/cfg_${cfg}_base.json
  else env $v python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/cfg_${cfg}_${v%%=*}.json 2> gpurun_out/cfg_${cfg}_${v%%=*}.err; f=gpurun_out/cfg_${cfg}_${v%%=*}.json; fi
  python - "$v" $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-28s value %.4g ms/step %.3f"%(sys.argv[1], d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["phase_ms"].items()})
except Exception as e: print(sys.argv[1],"ERR",e, open(sys.argv[2].replace(".json",".err")).read()[-300:])
PY
done
