#!/bin/bash
# Dry run of the multi-rank bench flow on a ONE-GPU box: two ranks share device 0 and the collectives go through the tests'
# shared-memory stand-in (tests/fake_rccl).  Checks rendezvous, unique-id broadcast, barriers, max-over-ranks timing and the
# rank-0 JSON line; the throughput it prints is meaningless (host-side all-reduce, time-sliced GPU).
set -e
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 -shared -fPIC -o /tmp/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp -lrt
PPO_RCCL_LIBRARY=/tmp/libfake_rccl.so python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --config ${1:-cfg4}
