"""ctypes binding of libppo_host.so: the C++ host layer (Env stack, Runner, PPO2) above the libppo_hip C-ABI."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HostArgs(C.Structure):
    _fields_ = [("n_envs", C.c_int), ("n_steps", C.c_int), ("n_hidden", C.c_int), ("hidden", C.c_int * 8),
                ("nminibatches", C.c_int), ("noptepochs", C.c_int), ("n_updates", C.c_int),
                ("lr", C.c_float), ("cliprange", C.c_float), ("gamma", C.c_float), ("lam", C.c_float),
                ("seeded_env", C.c_int), ("device", C.c_int), ("max_workers", C.c_int), ("reference_loop", C.c_int),
                ("norm_obs", C.c_int), ("norm_reward", C.c_int), ("seed", C.c_ulonglong), ("obs_dim", C.c_int), ("act_dim", C.c_int)]


class HostResult(C.Structure):
    _fields_ = [("env_steps_per_s", C.c_double), ("collect_ms", C.c_double), ("update_ms", C.c_double),
                ("losses", C.c_float * 5), ("fps_last", C.c_int), ("error", C.c_char * 256),
                ("obs_count", C.c_double), ("ret_count", C.c_double),
                ("phase_env_ms", C.c_double), ("phase_act_ms", C.c_double), ("phase_observe_ms", C.c_double),
                ("pool_workers", C.c_int), ("pool_chunk", C.c_int), ("pool_active", C.c_int)]


def load_host_library(build=True):
    global _LIB
    if _LIB is None:
        so = os.path.join(_PKG, "libppo_host.so")
        if build and os.path.exists("/opt/rocm/bin/hipcc"):
            from . import build as _b
            _b.build_hip()
            so = _b.build_host()
        C.CDLL(os.path.join(_PKG, "libppo_hip.so"), mode=C.RTLD_GLOBAL)
        _LIB = C.CDLL(so)
    return _LIB


def learn(n_envs, n_steps, hidden, n_updates, nminibatches=32, noptepochs=10, lr=3.93141e-4, cliprange=0.161023, gamma=0.99,
          lam=0.95, seeded_env=True, device=-1, max_workers=0, reference_loop=False, norm_obs=True, norm_reward=True, seed=0, obs_dim=18, act_dim=18):
    lib = load_host_library()
    a = HostArgs()
    a.n_envs, a.n_steps, a.n_hidden = n_envs, n_steps, len(hidden)
    for i, h in enumerate(hidden):
        a.hidden[i] = h
    a.nminibatches, a.noptepochs, a.n_updates = nminibatches, noptepochs, n_updates
    a.lr, a.cliprange, a.gamma, a.lam = lr, cliprange, gamma, lam
    a.seeded_env, a.device, a.max_workers, a.reference_loop = int(seeded_env), device, max_workers, int(reference_loop)
    a.norm_obs, a.norm_reward, a.seed = int(norm_obs), int(norm_reward), seed
    a.obs_dim, a.act_dim = obs_dim, act_dim
    r = HostResult()
    if lib.ppo_host_learn(C.byref(a), C.byref(r)) != 0:
        raise RuntimeError(r.error.decode())
    return {"env_steps_per_s": r.env_steps_per_s, "collect_ms": r.collect_ms, "update_ms": r.update_ms,
            "losses": [float(x) for x in r.losses], "fps_last": r.fps_last, "obs_count": r.obs_count, "ret_count": r.ret_count,
            "phase_ms": {"env_step": r.phase_env_ms, "act_kernel_d2h_sync": r.phase_act_ms, "observe_pack_h2d_enqueue": r.phase_observe_ms},
            "vec_env_pool": {"workers": r.pool_workers, "chunk": r.pool_chunk, "active_in_last_step": r.pool_active}}


class HostExplicit(C.Structure):
    _fields_ = [("theta_in", C.c_void_p), ("noise", C.c_void_p), ("perms", C.c_void_p), ("losses_out", C.c_void_p), ("theta_out", C.c_void_p),
                ("obs_mean", C.c_void_p), ("obs_var", C.c_void_p), ("obs_count", C.c_void_p),
                ("ret_mean", C.c_void_p), ("ret_var", C.c_void_p), ("ret_count", C.c_void_p), ("reward_curve", C.c_void_p)]


def learn_explicit(n_envs, n_steps, hidden, theta, noise, perms, nminibatches, lr=3.93141e-4, cliprange=0.161023, gamma=0.99, lam=0.95,
                   reference_loop=False, device=-1, obs_dim=18, act_dim=18):
    """PPO2::learn for perms.shape[0] updates with explicit weights, exploration noise [U,T,E,A] and epoch permutations [U,epochs,B]
    on SeededEnvMock x n_envs behind VecEnv + EnvNormalize.  Returns per-update mean losses [U,5], final weights, obs_rms, ret_rms."""
    import numpy as np
    lib = load_host_library()
    U, epochs, B = perms.shape
    assert noise.shape[:3] == (U, n_steps, n_envs) and B == n_envs * n_steps
    theta = np.ascontiguousarray(theta, np.float32); noise = np.ascontiguousarray(noise, np.float32); perms = np.ascontiguousarray(perms, np.int32)
    a = HostArgs()
    a.n_envs, a.n_steps, a.n_hidden = n_envs, n_steps, len(hidden)
    for i, h in enumerate(hidden):
        a.hidden[i] = h
    a.nminibatches, a.noptepochs, a.n_updates = nminibatches, epochs, U
    a.lr, a.cliprange, a.gamma, a.lam = lr, cliprange, gamma, lam
    a.seeded_env, a.device, a.max_workers, a.reference_loop = 1, device, 0, int(reference_loop)
    a.norm_obs, a.norm_reward, a.seed = 1, 1, 0
    a.obs_dim, a.act_dim = obs_dim, act_dim
    assert noise.shape[3] == act_dim
    out = {"losses": np.zeros((U, 5), np.float32), "theta": np.zeros(theta.size, np.float32),
           "obs_mean": np.zeros(obs_dim, np.float32), "obs_var": np.zeros(obs_dim, np.float32), "obs_count": np.zeros(1, np.float64),
           "ret_mean": np.zeros(1, np.float32), "ret_var": np.zeros(1, np.float32), "ret_count": np.zeros(1, np.float64)}
    x = HostExplicit(theta.ctypes.data, noise.ctypes.data, perms.ctypes.data, out["losses"].ctypes.data, out["theta"].ctypes.data,
                     out["obs_mean"].ctypes.data, out["obs_var"].ctypes.data, out["obs_count"].ctypes.data,
                     out["ret_mean"].ctypes.data, out["ret_var"].ctypes.data, out["ret_count"].ctypes.data, None)
    r = HostResult()
    if lib.ppo_host_learn_explicit(C.byref(a), C.byref(x), C.byref(r)) != 0:
        raise RuntimeError(r.error.decode())
    return out


def learn_curve(n_envs, n_steps, hidden, n_updates, nminibatches, noptepochs, lr, cliprange, gamma=0.99, lam=0.95, seed=0, reference_loop=False, device=-1,
                obs_dim=18, act_dim=18):
    """PPO2::learn on TargetEnv x n_envs (a learnable task, host/env/env_mock.hpp) behind VecEnv + EnvNormalize with the library's own exploration noise and shuffles:
    returns the mean un-normalised reward of every update's rollout [n_updates], the per-update mean losses and the final weights."""
    import numpy as np
    lib = load_host_library()
    a = HostArgs()
    a.n_envs, a.n_steps, a.n_hidden = n_envs, n_steps, len(hidden)
    for i, h in enumerate(hidden):
        a.hidden[i] = h
    a.nminibatches, a.noptepochs, a.n_updates = nminibatches, noptepochs, n_updates
    a.lr, a.cliprange, a.gamma, a.lam = lr, cliprange, gamma, lam
    a.seeded_env, a.device, a.max_workers, a.reference_loop = 2, device, 0, int(reference_loop)
    a.norm_obs, a.norm_reward, a.seed = 1, 1, seed
    a.obs_dim, a.act_dim = obs_dim, act_dim
    out = {"losses": np.zeros((n_updates, 5), np.float32), "reward_curve": np.zeros(n_updates, np.float32)}
    x = HostExplicit(None, None, None, out["losses"].ctypes.data, None, None, None, None, None, None, None, out["reward_curve"].ctypes.data)
    r = HostResult()
    if lib.ppo_host_learn_explicit(C.byref(a), C.byref(x), C.byref(r)) != 0:
        raise RuntimeError(r.error.decode())
    out["env_steps_per_s"] = r.env_steps_per_s
    return out
