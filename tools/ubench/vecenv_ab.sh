g++ -O2 -std=c++17 -pthread -I ppo_cpp_amd/host -o gpurun_out/vecenv_bench tools/ubench/vecenv_bench.cpp || exit 1
nproc
for gap in 0 100; do for n in 64 256 1024 4096; do for spin in 250 0; do echo -n "spin $spin: "; PPO_VECENV_SPIN_US=$spin gpurun_out/vecenv_bench $n 200 0 $gap | tail -1; done; done; done
for spin in 250 0; do PPO_VECENV_SPIN_US=$spin python - <<'PY'
import os, json
from ppo_cpp_amd import hostapi
for i in range(2):
    r = hostapi.learn(4096, 16, [256, 256], 4)
    print("spin", os.environ["PPO_VECENV_SPIN_US"], {k: r[k] for k in r if k in ("collect_ms", "collect_phase_ms", "vec_env_pool", "env_steps_per_s")})
PY
done
