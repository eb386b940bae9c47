out=gpurun_out/r05_a; mkdir -p $out; export TMPDIR=/tmp
pmc() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$n -- python3 bench.py --config cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/pmc_$n.json 2> $out/pmc_$n.err || tail -3 $out/pmc_$n.err
  cp $(find $out/pmc_$n -name "*counter_collection.csv" | head -1) $out/pmc_$n.csv; rm -rf $out/pmc_$n
  python tools/pmc_summary.py $out/pmc_${n}_summary.json "cfg3, one --pmc pass: $* ($NOTE)" $out/pmc_$n.csv; rm -f $out/pmc_$n.csv
}
export PPO_HIP_FUSE_AB=1
NOTE="fused launch, write-through phase A" pmc l2_fused_wt TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
export PPO_HIP_EXTRA_FLAGS="-DT8_WT=0 -DFAB_WT=0"; python -m ppo_cpp_amd.build --force >/dev/null 2>&1
NOTE="fused launch, plain phase-A stores (timing experiment)" pmc l2_fused_plain TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - <<PY
import json
for n in ("l2_fused_wt","l2_fused_plain"):
    d=json.load(open("$out/pmc_%s_summary.json"%n))["kernels"]
    for k,v in d.items():
        if "fused" in k or "adam" in k: print(n,k,v)
PY
