cd /tmp && export TMPDIR=/tmp
PPO_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_sequence.py $(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1) 100
