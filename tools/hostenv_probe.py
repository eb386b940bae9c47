import sys; sys.path.insert(0, '.')
from ppo_cpp_amd import hostapi
r = hostapi.learn(4096, 16, [256,256], n_updates=6, nminibatches=32, noptepochs=10)
print({k: r[k] for k in ("env_steps_per_s","collect_ms","update_ms","phase_ms")})
r = hostapi.learn(1, 2048, [64,64], n_updates=5, nminibatches=32, noptepochs=10)
print({k: r[k] for k in ("env_steps_per_s","collect_ms","update_ms","phase_ms")})
