out=gpurun_out/r04_q; mkdir -p $out; export TMPDIR=/tmp
python -m pytest tests/test_hip_parity.py tests/test_other_shapes.py -q -m gpu -x 2>&1 | grep -E "passed|failed" > $out/tests.txt; cat $out/tests.txt
rocprofv3 --list-avail 2>/dev/null | grep -oE "^\s*Name\s*:\s*\S+|Counter_Name\s*:\s*\S+" | sed 's/.*:\s*//' | sort -u > $out/counters.txt; wc -l $out/counters.txt
grep -E "^(TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_READ_sum|TCP_TCC_READ_REQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_TCC_READ_REQ_LATENCY_sum|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_LDS_ADDR_CONFLICT|GRBM_GUI_ACTIVE|SQ_INSTS_VMEM_RD|SQ_INSTS_LDS|SQ_WAIT_INST_LDS|SQ_INST_CYCLES_VMEM|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|SQ_BUSY_CYCLES|SQ_WAVES)$" $out/counters.txt | tr '\n' ' '
pmc() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$n -- python3 bench.py --config cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/pmc_$n.json 2> $out/pmc_$n.err || tail -3 $out/pmc_$n.err
  cp $(find $out/pmc_$n -name "*counter_collection.csv" | head -1) $out/pmc_$n.csv; rm -rf $out/pmc_$n
  python tools/pmc_summary.py $out/pmc_${n}_summary.json "cfg3, one --pmc pass: $*" $out/pmc_$n.csv; rm -f $out/pmc_$n.csv
}
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
pmc clk GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
ls $out
