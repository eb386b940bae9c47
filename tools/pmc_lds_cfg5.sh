#!/bin/bash
# LDS counters of the bf16 GEMM kernels at configs[4] (cfg5): how busy the LDS is under the k loop (the loop is LDS-byte bound).  One --pmc pass, eager launches
# (rocprofv3 crashes inside hipGraphLaunch of this configuration's update graph).  usage (GPU box): tools/pmc_lds_cfg5.sh <tag>
tag=${1:-r06_lds}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp PPO_HIP_NO_GRAPH=1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc -- python3 bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/pmc.json 2> $out/pmc.err || tail -3 $out/pmc.err
cp $(find $out/pmc -name "*counter_collection.csv" | head -1) $out/pmc_lds_cfg5.csv; rm -rf $out/pmc
python tools/pmc_summary.py $out/pmc_lds_cfg5.json "one --pmc pass, cfg5 (bf16), eager launches: LDS counters" $out/pmc_lds_cfg5.csv
python - $out/pmc_lds_cfg5.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["kernels"]
for k,v in d.items():
    if "gemm" in k or "reduce_adam" in k:
        print(k[:44], {n: v.get(n) for n in ("SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_LDS","SQ_ACTIVE_INST_LDS","SQ_BUSY_CU_CYCLES","SQ_WAVE_CYCLES")})
PY
