#!/bin/bash
# usage (GPU box): tools/variant_bench.sh "<extra hipcc -D flags>"  -> rebuilds libppo_hip.so with the flags, runs bench.py, prints kernel times
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $1 -o ppo_cpp_amd/libppo_hip.so ppo_cpp_amd/csrc/ppo_hip.hip -ldl || exit 1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'value %.4g ms/step %.2f' % (d['value'], d['ms_per_step'])); print('   ', {k: round(v['avg_us'],2) for k,v in d['kernels'].items()})"
