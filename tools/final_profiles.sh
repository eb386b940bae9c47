#!/bin/bash
# usage (GPU box, via gpurun): tools/final_profiles.sh <tag>  -> bench lines, rocprofv3 kernel stats and PMC passes of the
# configurations quoted in DESIGN.md, under gpurun_out/<tag>/
tag=${1:-r06_final}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 > $out/bench_cfg3.json 2> $out/bench_cfg3.err || tail -3 $out/bench_cfg3.err
for cfg in cfg2o36 cfg4o36; do python bench.py --config $cfg --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err; done
for cfg in cfg3 cfg3o36 cfg2 cfg4 cfg5; do
  [ $cfg != cfg3 ] && python bench.py --config $cfg --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  # (rocprofv3 segfaults inside hipGraphLaunch of the 4.2 k-node update graph of cfg5 on this image: eager launches for that one, same kernels)
  [ $cfg = cfg5 ] && export PPO_HIP_NO_GRAPH=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/bench_prof_$cfg.json 2> $out/prof_$cfg.err
  unset PPO_HIP_NO_GRAPH
  cp $(find $out/prof_$cfg -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$cfg.csv; rm -rf $out/prof_$cfg
done
pmc() { # name config counters...
  n=$1; cfg=$2; shift 2
  [ $cfg = cfg5 ] && export PPO_HIP_NO_GRAPH=1 || unset PPO_HIP_NO_GRAPH
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$n -- python3 bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/pmc_$n.json 2> $out/pmc_$n.err || tail -3 $out/pmc_$n.err
  cp $(find $out/pmc_$n -name "*counter_collection.csv" | head -1) $out/pmc_$n.csv; rm -rf $out/pmc_$n
}
pmc fetch_cfg3 cfg3 FETCH_SIZE
pmc write_cfg3 cfg3 WRITE_SIZE
pmc sq_cfg3 cfg3 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq_cfg5 cfg5 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc fetch_cfg5 cfg5 FETCH_SIZE
pmc write_cfg5 cfg5 WRITE_SIZE
unset PPO_HIP_NO_GRAPH
python tools/pmc_summary.py $out/hbm_traffic_cfg3.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --config cfg3 --steps 1 --warmup 1; KB per launch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports half of wide streaming reads, MI355X_MICROARCH.md)" $out/pmc_fetch_cfg3.csv $out/pmc_write_cfg3.csv
python tools/pmc_summary.py $out/pmc_sq_cfg3.json "one --pmc pass, cfg3; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles" $out/pmc_sq_cfg3.csv
python tools/pmc_summary.py $out/pmc_sq_cfg5.json "one --pmc pass, cfg5 (bf16)" $out/pmc_sq_cfg5.csv
python tools/pmc_summary.py $out/hbm_traffic_cfg5.json "FETCH_SIZE / WRITE_SIZE passes, cfg5 (bf16)" $out/pmc_fetch_cfg5.csv $out/pmc_write_cfg5.csv
ls $out | head -60
