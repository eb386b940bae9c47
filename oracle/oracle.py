"""ctypes binding of oracle/libppo_oracle.so (the C restatement in ppo_oracle.c).

TEST INFRASTRUCTURE - pinned to the reference graph as executed by oracle/graph_interp.py; no TensorFlow-produced vector
exists, so parity is "unpinned" at the TensorFlow boundary in that strict sense (see ppo_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_LAYERS = 8


class OrcCfg(C.Structure):
    _fields_ = [("obs_dim", C.c_int), ("act_dim", C.c_int), ("n_hidden", C.c_int),
                ("hidden", C.c_int * MAX_LAYERS), ("ent_coef", C.c_float), ("vf_coef", C.c_float),
                ("max_grad_norm", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float),
                ("adam_eps", C.c_float)]


class OrcRollout(C.Structure):
    _fields_ = [("E", C.c_int), ("T", C.c_int)] + [(n, C.POINTER(C.c_float)) for n in
                ("obs", "actions", "values", "neglogp", "dones", "rewards", "returns")]


class OrcRStats(C.Structure):
    _fields_ = [("dim", C.c_int), ("count", C.c_double), ("mean", C.POINTER(C.c_float)),
                ("var", C.POINTER(C.c_float))]


def build(force=False):
    so = os.path.join(_HERE, "libppo_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("ppo_oracle.c", "ppo_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libppo_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_clip_by_global_norm.restype = C.c_float
        _LIB.orc_train_step.restype = C.c_float
        _LIB.orc_tensor_info.restype = C.c_char_p
        _LIB.orc_hash.restype = C.c_uint32
    return _LIB


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# hyper-parameters baked into the reference graph G (tests/golden/g45_init.npz const_*):
G_ENT_COEF = 0.0007160293171182275
G_VF_COEF = 0.5
G_MAX_GRAD_NORM = 0.5
G_BETA1 = 0.8999999761581421
G_BETA2 = 0.9990000128746033
G_EPS = 9.999999747378752e-06


class Oracle:
    """Stateful convenience wrapper: weights + Adam slots + beta powers, mirroring the TF session state."""

    def __init__(self, obs_dim, act_dim, hidden, ent_coef=G_ENT_COEF, vf_coef=G_VF_COEF,
                 max_grad_norm=G_MAX_GRAD_NORM, beta1=G_BETA1, beta2=G_BETA2, eps=G_EPS):
        self.cfg = OrcCfg()
        self.cfg.obs_dim, self.cfg.act_dim, self.cfg.n_hidden = obs_dim, act_dim, len(hidden)
        for i, h in enumerate(hidden):
            self.cfg.hidden[i] = h
        self.cfg.ent_coef, self.cfg.vf_coef, self.cfg.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.cfg.adam_beta1, self.cfg.adam_beta2, self.cfg.adam_eps = beta1, beta2, eps
        self.O, self.A, self.hidden = obs_dim, act_dim, list(hidden)
        self.L = lib()
        self.P = self.L.orc_num_params(C.byref(self.cfg))
        self.theta = np.zeros(self.P, np.float32)
        self.m = np.zeros(self.P, np.float32)
        self.v = np.zeros(self.P, np.float32)
        self.pow = np.array([beta1, beta2], np.float32)      # G:25426, 25579: powers start at beta
        self.tensors = []
        for i in range(self.L.orc_num_tensors(C.byref(self.cfg))):
            off, r, c = C.c_int(), C.c_int(), C.c_int()
            name = self.L.orc_tensor_info(C.byref(self.cfg), i, C.byref(off), C.byref(r), C.byref(c)).decode()
            shape = (r.value, c.value) if c.value else (r.value,)
            self.tensors.append((name, off.value, shape))

    # -- parameters ---------------------------------------------------------------------------
    def tensor(self, name, arr=None):
        for n, off, shape in self.tensors:
            if n == name:
                cnt = int(np.prod(shape))
                return (arr if arr is not None else self.theta)[off:off + cnt].reshape(shape)
        raise KeyError(name)

    def named(self, arr=None):
        return {n: self.tensor(n, arr) for n, _, _ in self.tensors}

    def set_tensors(self, d):
        """d: name -> array (names like 'pi_fc0/w'); extra keys (q/w, q/b) are ignored."""
        for n, off, shape in self.tensors:
            self.theta[off:off + int(np.prod(shape))] = f32(d[n]).reshape(-1)

    def init_orthogonal(self, seed=0):
        """Same family as the initialisers in G (a16): orthogonal, gain sqrt2 hidden / 0.01 pi / 1.0 vf."""
        rng = np.random.RandomState(seed)
        for n, off, shape in self.tensors:
            cnt = int(np.prod(shape))
            if n.endswith("/w"):
                gain = 0.01 if n == "pi/w" else (1.0 if n == "vf/w" else np.sqrt(2.0))
                a = rng.normal(size=shape)
                u, _, vt = np.linalg.svd(a, full_matrices=False)
                q = u if u.shape == shape else vt
                self.theta[off:off + cnt] = (gain * q).astype(np.float32).reshape(-1)
            else:
                self.theta[off:off + cnt] = 0.0

    # -- act model ----------------------------------------------------------------------------
    def forward(self, obs):
        obs = f32(obs); n = obs.shape[0]
        mu = np.empty((n, self.A), np.float32); v = np.empty(n, np.float32)
        self.L.orc_forward(C.byref(self.cfg), _f(self.theta), _f(obs), n, _f(mu), _f(v))
        return mu, v

    def step(self, obs, noise):
        obs, noise = f32(obs), f32(noise); n = obs.shape[0]
        a = np.empty((n, self.A), np.float32); v = np.empty(n, np.float32); nlp = np.empty(n, np.float32)
        self.L.orc_step(C.byref(self.cfg), _f(self.theta), _f(obs), n, _f(noise), _f(a), _f(v), _f(nlp))
        return a, v, nlp

    # -- train --------------------------------------------------------------------------------
    def loss_grad(self, obs, actions, advs, returns, old_nlp, old_v, cliprange):
        args = [f32(x) for x in (obs, actions, advs, returns, old_nlp, old_v)]
        n = args[0].shape[0]
        losses = np.empty(5, np.float32); grad = np.empty(self.P, np.float32)
        self.L.orc_loss_grad(C.byref(self.cfg), _f(self.theta), *[_f(x) for x in args], n,
                             C.c_float(cliprange), _f(losses), _f(grad))
        return losses, grad

    def clip(self, grad):
        g = f32(grad).copy()
        norm = self.L.orc_clip_by_global_norm(C.byref(self.cfg), _f(g))
        return g, norm

    def adam(self, grad, lr):
        self.L.orc_adam(C.byref(self.cfg), _f(self.theta), _f(self.m), _f(self.v), _f(f32(grad)),
                        C.c_float(lr), _f(self.pow))

    def train_step(self, lr, cliprange, obs, actions, advs, returns, old_nlp, old_v):
        args = [f32(x) for x in (obs, actions, advs, returns, old_nlp, old_v)]
        n = args[0].shape[0]
        losses = np.empty(5, np.float32); grad = np.empty(self.P, np.float32)
        norm = self.L.orc_train_step(C.byref(self.cfg), _f(self.theta), _f(self.m), _f(self.v), _f(self.pow),
                                     C.c_float(lr), C.c_float(cliprange), *[_f(x) for x in args], n,
                                     _f(losses), _f(grad))
        return losses, norm, grad

    def update(self, ro, perms, nminibatches, lr, cliprange):
        """ro: dict of [T,E,...] float32 arrays (obs, actions, values, neglogp, returns); perms [epochs,B] int32."""
        T, E = ro["values"].shape
        keep = {k: f32(ro[k]) for k in ("obs", "actions", "values", "neglogp", "returns")}
        r = OrcRollout()
        r.E, r.T = E, T
        for k, a in keep.items():
            setattr(r, k, _f(a))
        perms = np.ascontiguousarray(perms, np.int32)
        epochs = perms.shape[0]
        rows = np.empty((epochs * nminibatches, 5), np.float32); mean = np.empty(5, np.float32)
        self.L.orc_update(C.byref(self.cfg), _f(self.theta), _f(self.m), _f(self.v), _f(self.pow), C.byref(r),
                          perms.ctypes.data_as(C.POINTER(C.c_int32)), epochs, nminibatches,
                          C.c_float(lr), C.c_float(cliprange), _f(rows), _f(mean))
        return rows, mean


# -- stateless host-side numerics ---------------------------------------------------------------
def adv_normalize(returns, values):
    r, v = f32(returns), f32(values)
    out = np.empty_like(r)
    lib().orc_adv_normalize(_f(r), _f(v), r.size, _f(out))
    return out


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    rw, va, dn, lv, ld = [f32(x) for x in (rewards, values, dones, last_values, last_dones)]
    T, E = rw.shape
    out = np.empty((T, E), np.float32)
    lib().orc_gae(_f(rw), _f(va), _f(dn), _f(lv), _f(ld), T, E, C.c_float(gamma), C.c_float(lam), _f(out))
    return out


class RunningStats:
    """common/running_statistics.hpp restated (mean 0, var 1, count 1e-6)."""

    def __init__(self, dim):
        self.mean = np.zeros(dim, np.float32)
        self.var = np.ones(dim, np.float32)
        self.s = OrcRStats()
        lib().orc_rstats_init(C.byref(self.s), dim, _f(self.mean), _f(self.var))

    @property
    def count(self):
        return self.s.count

    @count.setter
    def count(self, c):
        self.s.count = c

    def update(self, batch):
        b = f32(batch).reshape(-1, self.s.dim)
        lib().orc_rstats_update(C.byref(self.s), _f(b), b.shape[0])


class Normalizer:
    """env/env_normalize.hpp restated for [E,O] observations and [E] rewards."""

    def __init__(self, n_envs, obs_dim, gamma=0.99, clip_obs=10.0, clip_rew=10.0, eps=1e-8, training=True):
        self.obs_rms, self.ret_rms = RunningStats(obs_dim), RunningStats(1)
        self.ret = np.zeros(n_envs, np.float32)
        self.gamma, self.clip_obs, self.clip_rew, self.eps, self.training = gamma, clip_obs, clip_rew, eps, training

    def obs(self, x):
        x = f32(x); out = np.empty_like(x)
        lib().orc_normalize_obs(C.byref(self.obs_rms.s), _f(x), x.shape[0], int(self.training),
                                C.c_float(self.clip_obs), C.c_float(self.eps), _f(out))
        return out

    def reward(self, rew, dones):
        r, d = f32(rew).reshape(-1), f32(dones).reshape(-1); out = np.empty_like(r)
        lib().orc_normalize_reward(C.byref(self.ret_rms.s), _f(self.ret), _f(r), _f(d), r.size,
                                   int(self.training), C.c_float(self.gamma), C.c_float(self.clip_rew),
                                   C.c_float(self.eps), _f(out))
        return out


def seeded_env_step(seed, env0, n_envs, step, obs_dim):
    obs = np.empty((n_envs, obs_dim), np.float32); rew = np.empty(n_envs, np.float32); dn = np.empty(n_envs, np.float32)
    lib().orc_seeded_env_step(C.c_uint32(seed), env0, n_envs, C.c_uint32(step), obs_dim, _f(obs), _f(rew), _f(dn))
    return obs, rew, dn


def collect(orc, norm, seed, T, noise, gamma, lam, step0=0, state=None):
    """runner.hpp:56-157 restated over the seeded synthetic env.  noise [T,E,A].  Returns the rollout dict
    (time-major [T,E,...]), the carry state (obs, dones) for the next call and the bootstrap values."""
    E = norm.ret.size
    O, A = orc.O, orc.A
    if state is None:
        raw, _, _ = seeded_env_step(seed, 0, E, step0, O)     # reset() observation = counter value step0
        state = (norm.obs(raw), np.zeros(E, np.float32))       # Runner ctor: env.reset(), dones = 0 (runner.hpp:48-50)
    obs, dones = state
    ro = {k: np.empty((T, E) + s, np.float32) for k, s in
          (("obs", (O,)), ("actions", (A,)), ("values", ()), ("neglogp", ()), ("dones", ()), ("rewards", ()))}
    for t in range(T):
        ro["obs"][t] = obs
        a, v, nlp = orc.step(obs, noise[t])
        ro["actions"][t], ro["values"][t], ro["neglogp"][t], ro["dones"][t] = a, v, nlp, dones
        raw, rew, dones = seeded_env_step(seed, 0, E, step0 + t + 1, O)
        obs = norm.obs(raw)                                    # env_normalize.hpp:74
        ro["rewards"][t] = norm.reward(rew, dones)             # env_normalize.hpp:71,76-88
    _, last_v = orc.forward(obs)                               # runner.hpp:161-165
    ro["returns"] = gae(ro["rewards"], ro["values"], ro["dones"], last_v, dones, gamma, lam)
    return ro, (obs, dones), last_v
