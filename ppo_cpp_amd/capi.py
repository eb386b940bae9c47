"""ctypes binding of libppo_hip.so (include/ppo_hip.h).  Thin: every method is one C-ABI call."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_LAYERS = 8


class PPOHipError(RuntimeError):
    pass


class PPOConfig(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MAX_LAYERS), ("ent_coef", C.c_float), ("vf_coef", C.c_float),
                ("max_grad_norm", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float),
                ("adam_eps", C.c_float), ("device", C.c_int32), ("max_rows", C.c_int32), ("compute_dtype", C.c_int32)]


def load_library(build=True):
    """Loads ppo_cpp_amd/libppo_hip.so (building it in-tree when the sources are newer)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_PKG, "libppo_hip.so")
    alt = os.environ.get("PPO_HIP_LIBRARY")                  # a diagnostic build (tools/stamps*.py: -DPPO_STAMPS) kept OUTSIDE the package directory
    if alt:
        so = alt
    elif build and os.path.exists("/opt/rocm/bin/hipcc"):
        from . import build as _b
        so = _b.build_hip()
    if not os.path.exists(so):
        raise PPOHipError("libppo_hip.so is missing (%s): run `python -m ppo_cpp_amd.build`; there is no CPU fallback" % so)
    lib = C.CDLL(so)
    lib.ppo_last_error.restype = C.c_char_p
    lib.ppo_last_error.argtypes = [C.c_void_p]
    lib.ppo_destroy.argtypes = [C.c_void_p]
    lib.ppo_destroy.restype = None
    _LIB = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


class PPOHip:
    """One handle = one GPU + stream; mirrors the TF session state of the reference (weights, Adam slots, beta
    powers) plus the device-resident rollout and normaliser."""

    FIELDS = {"obs": 0, "actions": 1, "values": 2, "neglogp": 3, "dones": 4, "rewards": 5, "returns": 6}

    def __init__(self, obs_dim, act_dim, hidden, device=-1, **overrides):
        self.lib = load_library()
        cfg = PPOConfig()
        hid = (C.c_int32 * len(hidden))(*hidden)
        self.lib.ppo_config_default(C.byref(cfg), obs_dim, act_dim, len(hidden), hid)
        cfg.device = device
        for k, v in overrides.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.O, self.A, self.hidden = obs_dim, act_dim, list(hidden)
        h = C.c_void_p()
        if self.lib.ppo_create(C.byref(cfg), C.byref(h)) != 0:
            raise PPOHipError(self.lib.ppo_last_error(None).decode())
        self.h = h
        self.P = self.lib.ppo_num_params(self.h)
        self.tensors = []
        for i in range(self.lib.ppo_num_tensors(self.h)):
            name = C.create_string_buffer(32)
            r, c = C.c_int32(), C.c_int32()
            self._ck(self.lib.ppo_tensor_info(self.h, i, name, C.byref(r), C.byref(c)))
            self.tensors.append((name.value.decode(), (r.value, c.value) if c.value else (r.value,)))
        self.E = self.T = 0
        self.world, self._global_shuffle = 1, False

    def close(self):
        if getattr(self, "h", None):
            self.lib.ppo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise PPOHipError(self.lib.ppo_last_error(self.h).decode())

    # ---- variables --------------------------------------------------------------------------------
    def get_flat(self, which=0):
        out = np.empty(self.P, np.float32)
        self._ck(self.lib.ppo_get_flat(self.h, which, _fp(out), C.c_int64(self.P)))
        return out

    def set_flat(self, arr, which=0):
        a = _f32(arr, (self.P,))
        self._ck(self.lib.ppo_set_flat(self.h, which, _fp(a), C.c_int64(self.P)))

    def set_tensors(self, named):
        for i, (name, shape) in enumerate(self.tensors):
            a = _f32(named[name]).reshape(-1)
            self._ck(self.lib.ppo_set_tensor(self.h, 0, i, _fp(a), C.c_int64(a.size)))

    def get_tensor(self, name, which=0):
        for i, (n, shape) in enumerate(self.tensors):
            if n == name:
                out = np.empty(int(np.prod(shape)), np.float32)
                self._ck(self.lib.ppo_get_tensor(self.h, which, i, _fp(out), C.c_int64(out.size)))
                return out.reshape(shape)
        raise KeyError(name)

    def init_orthogonal(self, seed=0):
        self._ck(self.lib.ppo_init_orthogonal(self.h, C.c_uint64(seed)))

    def seed(self, seed):
        self._ck(self.lib.ppo_seed(self.h, C.c_uint64(seed)))

    def beta_powers(self):
        pw = np.empty(2, np.float32)
        self._ck(self.lib.ppo_get_beta_powers(self.h, _fp(pw)))
        return pw

    def set_beta_powers(self, pw):
        a = _f32(pw, (2,))
        self._ck(self.lib.ppo_set_beta_powers(self.h, _fp(a)))

    # ---- act model --------------------------------------------------------------------------------
    def step(self, obs, noise=None):
        obs = _f32(obs); n = obs.shape[0]
        a = np.empty((n, self.A), np.float32); v = np.empty(n, np.float32); nlp = np.empty(n, np.float32)
        nz = _f32(noise, (n, self.A)) if noise is not None else None
        self._ck(self.lib.ppo_step(self.h, _fp(obs), n, _fp(nz) if nz is not None else None, _fp(a), _fp(v), _fp(nlp)))
        return a, v, nlp

    def value(self, obs):
        obs = _f32(obs); n = obs.shape[0]
        v = np.empty(n, np.float32)
        self._ck(self.lib.ppo_value(self.h, _fp(obs), n, _fp(v)))
        return v

    def act_deterministic(self, obs):
        obs = _f32(obs); n = obs.shape[0]
        a = np.empty((n, self.A), np.float32)
        self._ck(self.lib.ppo_act_deterministic(self.h, _fp(obs), n, _fp(a)))
        return a

    # ---- train ------------------------------------------------------------------------------------
    def train_step(self, lr, cliprange, obs, actions, advs, returns, old_nlp, old_v):
        arrs = [_f32(x) for x in (obs, actions, advs, returns, old_nlp, old_v)]
        n = arrs[0].shape[0]
        losses = np.empty(5, np.float32)
        self._ck(self.lib.ppo_train_step(self.h, C.c_float(lr), C.c_float(cliprange), *[_fp(x) for x in arrs], n, _fp(losses)))
        return losses

    def last_grad(self):
        g = np.empty(self.P, np.float32); norm = C.c_float()
        self._ck(self.lib.ppo_get_last_grad(self.h, _fp(g), C.c_int64(self.P), C.byref(norm)))
        return g, norm.value

    def adv_normalize(self, returns, values):
        r, v = _f32(returns), _f32(values)
        out = np.empty_like(r)
        self._ck(self.lib.ppo_adv_normalize(self.h, _fp(r), _fp(v), r.size, _fp(out)))
        return out

    def gae(self, rewards, values, dones, last_values, last_dones, gamma, lam):
        rw, va, dn, lv, ld = [_f32(x) for x in (rewards, values, dones, last_values, last_dones)]
        T, E = rw.shape
        out = np.empty((T, E), np.float32)
        self._ck(self.lib.ppo_gae(self.h, _fp(rw), _fp(va), _fp(dn), _fp(lv), _fp(ld), T, E, C.c_float(gamma), C.c_float(lam), _fp(out)))
        return out

    # ---- normaliser -------------------------------------------------------------------------------
    def norm_init(self, n_envs, gamma=0.99, clip_obs=10.0, clip_rew=10.0, eps=1e-8):
        self._ck(self.lib.ppo_norm_init(self.h, n_envs, C.c_float(gamma), C.c_float(clip_obs), C.c_float(clip_rew), C.c_float(eps)))

    def norm_set_flags(self, norm_obs=True, norm_reward=True):
        self._ck(self.lib.ppo_norm_set_flags(self.h, int(norm_obs), int(norm_reward)))

    def norm_reset_returns(self):
        self._ck(self.lib.ppo_norm_reset_returns(self.h))

    def norm_obs(self, raw, training=True):
        x = _f32(raw); out = np.empty_like(x)
        self._ck(self.lib.ppo_norm_obs(self.h, _fp(x), x.shape[0], int(training), _fp(out)))
        return out

    def norm_reward(self, rew, dones, training=True):
        r, d = _f32(rew).reshape(-1), _f32(dones).reshape(-1); out = np.empty_like(r)
        self._ck(self.lib.ppo_norm_reward(self.h, _fp(r), _fp(d), r.size, int(training), _fp(out)))
        return out

    def norm_stats(self, which):
        dim = self.O if which == 0 else 1
        mean = np.empty(dim, np.float32); var = np.empty(dim, np.float32); cnt = C.c_double()
        self._ck(self.lib.ppo_norm_get_stats(self.h, which, _fp(mean), _fp(var), C.byref(cnt)))
        return mean, var, cnt.value

    def set_norm_stats(self, which, mean, var, count):
        m, v = _f32(mean).reshape(-1), _f32(var).reshape(-1)
        self._ck(self.lib.ppo_norm_set_stats(self.h, which, _fp(m), _fp(v), C.c_double(count)))

    # ---- rollout ----------------------------------------------------------------------------------
    def rollout_alloc(self, n_envs, n_steps):
        self._ck(self.lib.ppo_rollout_alloc(self.h, n_envs, n_steps))
        self.E, self.T = n_envs, n_steps

    def rollout_reset(self, raw_obs):
        x = _f32(raw_obs, (self.E, self.O))
        self._ck(self.lib.ppo_rollout_reset(self.h, _fp(x)))

    def rollout_act(self, t, noise=None):
        out = np.empty((self.E, self.A), np.float32)
        nz = _f32(noise, (self.E, self.A)) if noise is not None else None
        self._ck(self.lib.ppo_rollout_act(self.h, t, _fp(nz) if nz is not None else None, _fp(out)))
        return out

    def rollout_observe(self, t, raw_obs, raw_rew, dones):
        o, r, d = _f32(raw_obs, (self.E, self.O)), _f32(raw_rew).reshape(-1), _f32(dones).reshape(-1)
        self._ck(self.lib.ppo_rollout_observe(self.h, t, _fp(o), _fp(r), _fp(d)))

    def rollout_finish(self, gamma, lam):
        self._ck(self.lib.ppo_rollout_finish(self.h, C.c_float(gamma), C.c_float(lam)))

    def collect_synthetic(self, seed, gamma, lam, noise=None, env0=0, step0=0, first=True):
        nz = _f32(noise, (self.T, self.E, self.A)) if noise is not None else None
        self._ck(self.lib.ppo_collect_synthetic(self.h, C.c_uint32(seed), env0, C.c_uint32(step0), int(first),
                                                _fp(nz) if nz is not None else None, C.c_float(gamma), C.c_float(lam)))

    def rollout_get(self, field):
        shape = {"obs": (self.T, self.E, self.O), "actions": (self.T, self.E, self.A)}.get(field, (self.T, self.E))
        out = np.empty(shape, np.float32)
        self._ck(self.lib.ppo_rollout_download(self.h, self.FIELDS[field], _fp(out), C.c_int64(out.size)))
        return out

    def rollout_set(self, field, arr):
        a = _f32(arr)
        self._ck(self.lib.ppo_rollout_upload(self.h, self.FIELDS[field], _fp(a), C.c_int64(a.size)))

    def update(self, lr, cliprange, noptepochs, nminibatches, perms=None, seed=0, want_rows=True):
        rows = np.empty((noptepochs * nminibatches, 5), np.float32) if want_rows else None
        mean = np.empty(5, np.float32)
        pp = None
        if perms is not None:
            perms = np.ascontiguousarray(perms, np.int32)
            cols = self.E * self.T * (self.world if self._global_shuffle else 1)      # (B * world columns under ppo_dist_global_shuffle)
            assert perms.shape == (noptepochs, cols), (perms.shape, (noptepochs, cols))
            pp = perms.ctypes.data_as(C.POINTER(C.c_int32))
        self._ck(self.lib.ppo_update(self.h, C.c_float(lr), C.c_float(cliprange), noptepochs, nminibatches, pp, C.c_uint64(seed),
                                     _fp(rows) if rows is not None else None, _fp(mean)))
        return rows, mean

    # ---- dist / measurement -----------------------------------------------------------------------
    @staticmethod
    def dist_unique_id():
        lib = load_library()
        uid = C.create_string_buffer(128)
        if lib.ppo_dist_unique_id(uid) != 0:
            raise PPOHipError(lib.ppo_last_error(None).decode())
        return uid.raw

    def dist_init(self, world, rank, uid):
        assert len(uid) == 128
        self._ck(self.lib.ppo_dist_init(self.h, world, rank, C.create_string_buffer(uid, 128)))
        self.world = world

    def dist_info(self):
        """{comm_nranks (ncclCommCount), device (HIP ordinal), pci_bus_id, library (path of the collective library loaded)}"""
        n, d = C.c_int32(), C.c_int32()
        pci, lib = C.create_string_buffer(32), C.create_string_buffer(256)
        self._ck(self.lib.ppo_dist_info(self.h, C.byref(n), C.byref(d), pci, lib))
        return {"comm_nranks": n.value, "device": d.value, "pci_bus_id": pci.value.decode(), "library": lib.value.decode()}

    def dist_graph_collectives(self):
        return bool(self.lib.ppo_dist_graph_collectives(self.h))

    def dist_peer_export(self):
        """64-byte IPC handle of this rank's gather region (one-shot peer all-reduce); all-gather them and pass to dist_peer_attach"""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.ppo_dist_peer_export(self.h, buf))
        return buf.raw

    def dist_peer_attach(self, handles):
        """handles: world x 64 bytes in rank order.  Collective.  Returns True when the peer path is now in use."""
        blob = b"".join(handles) if not isinstance(handles, (bytes, bytearray)) else bytes(handles)
        self._ck(self.lib.ppo_dist_peer_attach(self.h, C.create_string_buffer(blob, len(blob))))
        return self.dist_peer_active()

    def dist_peer_active(self):
        return bool(self.lib.ppo_dist_peer_active(self.h))

    def dist_global_shuffle(self, on=True):
        self._ck(self.lib.ppo_dist_global_shuffle(self.h, int(on)))
        self._global_shuffle = bool(on) and self.world > 1

    def dist_bucketed(self, on=True):
        """bf16 path under a communicator: gradient buckets on a second stream (True, the default) / one all-reduce (False) / 2: buckets also at world 1 (measurement)"""
        self._ck(self.lib.ppo_dist_bucketed(self.h, int(on)))

    def dist_peer_enable(self, on=True):
        self._ck(self.lib.ppo_dist_peer_enable(self.h, int(on)))

    def prof_enable(self, on=True):
        self._ck(self.lib.ppo_prof_enable(self.h, int(on)))

    def prof_read(self):
        names = ((C.c_char * 32) * 16)(); ms = (C.c_double * 16)(); cnt = (C.c_int64 * 16)()
        n = self.lib.ppo_prof_read(self.h, 16, names, ms, cnt)
        return {names[i].value.decode(): (ms[i], cnt[i]) for i in range(n)}

    def kernel_counts(self):
        """{kernel variant: times enqueued since creation} -- which of the shape-selected kernels the calls so far took"""
        names = ((C.c_char * 32) * 32)(); cnt = (C.c_int64 * 32)()
        n = self.lib.ppo_kernel_counts(self.h, 32, names, cnt)
        return {names[i].value.decode(): cnt[i] for i in range(n)}

    def debug_buffer(self, name):
        """a raw device buffer by name, padding included, as uint32 words (include/ppo_hip.h, ppo_debug_buffer); empty when this shape does not use it"""
        cnt = C.c_int64(0)
        self._ck(self.lib.ppo_debug_buffer(self.h, name.encode(), None, C.c_int64(0), C.byref(cnt)))
        out = np.empty(cnt.value, np.uint32)
        if cnt.value:
            self._ck(self.lib.ppo_debug_buffer(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(cnt.value), C.byref(cnt)))
        return out

    def debug_graph_nodes(self):
        """{kernel, memset, memcpy, other: count} of the update's captured hipGraph, or None when the handle holds none (include/ppo_hip.h, ppo_debug_graph_nodes)"""
        c = (C.c_int32 * 4)()
        if self.lib.ppo_debug_graph_nodes(self.h, c) != 0:
            return None
        return dict(zip(("kernel", "memset", "memcpy", "other"), [int(x) for x in c]))

    def debug_raise_chain_error(self):
        self._ck(self.lib.ppo_debug_raise_chain_error(self.h))

    def debug_poison_lds(self, word=0x7FC0DEAD):
        """leave `word` (default: a NaN pattern) in every LDS word of every CU (include/ppo_hip.h, ppo_debug_poison_lds)"""
        self._ck(self.lib.ppo_debug_poison_lds(self.h, C.c_uint32(word)))

    def sync(self):
        self._ck(self.lib.ppo_sync(self.h))
