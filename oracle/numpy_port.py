"""A VECTORISED CPU port of the hot path: the same statements as oracle/ppo_oracle.c (which restates the reference graph `G` and the
host loop), written as NumPy array expressions over BLAS sgemm -- what a competent CPU implementation of the reference's TF-Eigen
path costs, as opposed to the scalar double-accumulating loops of the C restatement.

TEST INFRASTRUCTURE (like everything under oracle/): imported only by tests/ (tests/test_oracle.py checks it against the C oracle)
and by bench.py's `cpu_baseline` leg, which times it on 1 thread and on all granted host cores.  Never by the product.

Covered per train step (reference ppo2/ppo2.hpp:430-468 = one Session::Run of `ppo2/_train`): both towers' forward (MatMul + BiasAdd +
Tanh, G:6889-9187), neglogp / entropy / clipped surrogate / clipped value loss (G:9210-11446), their gradients with TF's tie rules
(G:12609-22656), TanhGrad + the MatMul_grad pairs (G:11773-23699), clip_by_global_norm (G:23738-25392) and ApplyAdam over all
parameters (G:25426-31383).  Per env step (ppo2/runner.hpp:75-116): the act model forward, sampling and neglogp (G:1859-6866).
"""
import numpy as np

HALF_LOG_2PI = np.float32(0.9189385175704956)
HALF_LOG_2PIE = np.float32(1.4189385175704956)
F = np.float32


class NumpyPPO:
    """Holds views into an oracle.Oracle's flat theta / m / v / pow (same tensor order and layout), so the two can be compared directly."""

    def __init__(self, orc):
        self.orc = orc
        self.L = len(orc.hidden)
        c = orc.cfg
        self.ent, self.vfc, self.maxn = F(c.ent_coef), F(c.vf_coef), F(c.max_grad_norm)
        self.b1, self.b2, self.eps = F(c.adam_beta1), F(c.adam_beta2), F(c.adam_eps)
        self.grad = np.zeros_like(orc.theta)

    def _t(self, name, arr=None):
        return self.orc.tensor(name, arr)

    # ---- act model ------------------------------------------------------------------------------------------------------------------
    def _tower(self, x, pre, keep):
        hs = [x]
        h = x
        for l in range(self.L):
            h = np.tanh(h @ self._t("%s_fc%d/w" % (pre, l)) + self._t("%s_fc%d/b" % (pre, l)))
            if keep:
                hs.append(h)
        return h, hs

    def forward(self, obs):
        hp, _ = self._tower(obs, "pi", False)
        hv, _ = self._tower(obs, "vf", False)
        return hp @ self._t("pi/w") + self._t("pi/b"), (hv @ self._t("vf/w")).reshape(-1) + self._t("vf/b")[0]

    def step(self, obs, noise):
        mu, v = self.forward(obs)
        logstd = self._t("pi/logstd").reshape(-1)
        sigma = np.exp(logstd)
        a = mu + sigma * noise
        z = (a - mu) / sigma
        nlp = F(0.5) * (z * z).sum(1, dtype=np.float32) + HALF_LOG_2PI * F(self.orc.A) + logstd.sum(dtype=np.float32)
        return a.astype(np.float32), v.astype(np.float32), nlp.astype(np.float32)

    # ---- train op -------------------------------------------------------------------------------------------------------------------
    def loss_grad(self, obs, actions, advs, returns, old_nlp, old_v, cr):
        n = obs.shape[0]
        A = self.orc.A
        cr = F(cr)
        g = F(1.0) / F(n)
        hp, hps = self._tower(obs, "pi", True)
        hv, hvs = self._tower(obs, "vf", True)
        w_mu, w_v = self._t("pi/w"), self._t("vf/w")
        mu = hp @ w_mu + self._t("pi/b")
        v = (hv @ w_v).reshape(-1) + self._t("vf/b")[0]
        logstd = self._t("pi/logstd").reshape(-1)
        sigma = np.exp(logstd)
        z = (actions - mu) / sigma
        nlp = F(0.5) * (z * z).sum(1, dtype=np.float32) + HALF_LOG_2PI * F(A) + logstd.sum(dtype=np.float32)
        ent = (logstd + HALF_LOG_2PIE).sum(dtype=np.float32)
        # value loss
        dvo = v - old_v
        vmin = np.minimum(dvo, cr)
        vclip = old_v + np.maximum(vmin, -cr)
        e1, e2 = v - returns, vclip - returns
        s1, s2 = e1 * e1, e2 * e2
        # policy loss
        lo, hi = F(1.0) - cr, F(1.0) + cr
        ratio = np.exp(old_nlp - nlp)
        rmin = np.minimum(ratio, hi)
        rclip = np.maximum(rmin, lo)
        m1, m2 = -advs * ratio, -advs * rclip
        dk = nlp - old_nlp
        losses = np.array([np.maximum(m1, m2).mean(dtype=np.float64), 0.5 * np.maximum(s1, s2).mean(dtype=np.float64), ent,
                           0.5 * (dk * dk).mean(dtype=np.float64), (np.abs(ratio - F(1.0)) > cr).mean(dtype=np.float64)], np.float32)
        # backward heads (TF tie rules: Maximum -> first argument iff x >= y)
        sel = (m1 >= m2).astype(np.float32)
        passp = ((rmin >= lo) & (ratio <= hi)).astype(np.float32)
        d_ratio = (-advs) * g * sel + (-advs) * g * (F(1.0) - sel) * passp
        d_nlp = -(d_ratio * ratio)
        dl = d_nlp[:, None] * (F(1.0) - z * z) - self.ent * g
        d_mu = d_nlp[:, None] * (-(z / sigma))
        gv = self.vfc * F(0.5) * g
        selv = (s1 >= s2).astype(np.float32)
        passv = ((vmin >= -cr) & (dvo <= cr)).astype(np.float32)
        d_v = gv * selv * (F(2.0) * e1) + gv * (F(1.0) - selv) * (F(2.0) * e2) * passv
        G = self.grad
        G[:] = 0
        self._t("pi/logstd", G)[:] = dl.sum(0, dtype=np.float32)
        self._t("pi/w", G)[:] = hp.T @ d_mu
        self._t("pi/b", G)[:] = d_mu.sum(0, dtype=np.float32)
        self._t("vf/w", G)[:] = hv.T @ d_v[:, None]
        self._t("vf/b", G)[:] = d_v.sum(dtype=np.float32)
        for pre, hs, dh in (("pi", hps, d_mu @ w_mu.T), ("vf", hvs, d_v[:, None] * w_v.reshape(1, -1))):
            for l in range(self.L - 1, -1, -1):
                y = hs[l + 1]
                dh = dh * (F(1.0) - y * y)                                       # TanhGrad
                self._t("%s_fc%d/w" % (pre, l), G)[:] = hs[l].T @ dh
                self._t("%s_fc%d/b" % (pre, l), G)[:] = dh.sum(0, dtype=np.float32)
                if l:
                    dh = dh @ self._t("%s_fc%d/w" % (pre, l)).T
        return losses, G

    def clip_adam(self, grad, lr):
        """clip_by_global_norm + ApplyAdam over the flat vectors, in place (two scratch vectors, no other temporaries)"""
        o = self.orc
        norm = np.float32(np.sqrt(np.dot(grad, grad)))                                   # sqrt(2 * sum of L2Loss): BLAS sdot
        scale = self.maxn * min(F(1.0) / norm, F(1.0) / self.maxn) if np.isfinite(norm) else F(np.nan)
        if not hasattr(self, "_t0"):
            self._t0 = np.empty_like(grad); self._t1 = np.empty_like(grad)
        gs, t = self._t0, self._t1
        np.multiply(grad, F(scale), out=gs)
        b1p, b2p = o.pow
        alpha = F(lr) * np.sqrt(F(1.0) - b2p) / (F(1.0) - b1p)
        np.subtract(gs, o.m, out=t); t *= (F(1.0) - self.b1); o.m += t                   # m += (g - m)(1 - b1)
        np.multiply(gs, gs, out=t); t -= o.v; t *= (F(1.0) - self.b2); o.v += t           # v += (g^2 - v)(1 - b2)
        np.sqrt(o.v, out=t); t += self.eps
        np.multiply(o.m, alpha, out=gs); gs /= t; o.theta -= gs                          # theta -= alpha m / (sqrt(v) + eps)
        o.pow[0] = b1p * self.b1
        o.pow[1] = b2p * self.b2
        return norm

    def train_step(self, lr, cr, obs, actions, advs, returns, old_nlp, old_v):
        losses, grad = self.loss_grad(obs, actions, advs, returns, old_nlp, old_v, cr)
        norm = self.clip_adam(grad, lr)
        return losses, norm


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    """Runner::set_returns (ppo2/runner.hpp:159-191), vectorised over the environments"""
    T, E = rewards.shape
    adv = np.empty((T, E), np.float32)
    last = np.zeros(E, np.float32)
    gamma, lam = F(gamma), F(lam)
    for t in range(T - 1, -1, -1):
        nonterm = F(1.0) - (last_dones if t == T - 1 else dones[t + 1])
        nextv = last_values if t == T - 1 else values[t + 1]
        delta = rewards[t] + gamma * nextv * nonterm - values[t]
        last = delta + gamma * lam * nonterm * last
        adv[t] = last
    return adv + values


def running_update(mean, var, count, batch):
    """RunningStatistics::update (common/running_statistics.hpp:26-104) on a [rows, D] batch; returns (mean, var, count)"""
    nb = batch.shape[0]
    bmean = batch.mean(0, dtype=np.float32)
    bvar = ((batch - bmean) ** 2).mean(0, dtype=np.float32)
    tot = count + nb
    delta = bmean - mean
    mean1 = mean + delta * F(nb) / F(tot)
    M2 = var * F(count) + bvar * F(nb) + delta * delta * F(count) * F(nb) / F(tot)
    return mean1, M2 / F(tot), tot
