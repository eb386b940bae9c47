"""Independent restatement of the reference graph's arithmetic in torch (CPU, float64, autograd).

TEST INFRASTRUCTURE.  Used to pin the C oracle (ppo_oracle.c): the forward / loss are written from the
formula sheet (recovered from G) with torch ops, and the backward comes from autograd
instead of the hand-derived TF gradient graph, so it shares no code with the C restatement.
"""
import numpy as np
import torch

HALF_LOG_2PI = 0.5 * np.log(2.0 * np.pi)
HALF_LOG_2PIE = 0.5 * np.log(2.0 * np.pi * np.e)


def as_params(named, dtype=torch.float64):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in named.items()}


def forward(p, obs, n_hidden):
    hp = hv = obs
    for l in range(n_hidden):
        hp = torch.tanh(hp @ p["pi_fc%d/w" % l] + p["pi_fc%d/b" % l])
        hv = torch.tanh(hv @ p["vf_fc%d/w" % l] + p["vf_fc%d/b" % l])
    mu = hp @ p["pi/w"] + p["pi/b"]
    v = (hv @ p["vf/w"] + p["vf/b"])[:, 0]
    logstd = mu * 0.0 + p["pi/logstd"].reshape(1, -1)
    return mu, v, logstd


def neglogp(a, mu, logstd):
    return 0.5 * (((a - mu) / torch.exp(logstd)) ** 2).sum(-1) + HALF_LOG_2PI * a.shape[-1] + logstd.sum(-1)


def losses(p, n_hidden, obs, act, adv, ret, old_nlp, old_v, cr, ent_coef, vf_coef):
    mu, v, logstd = forward(p, obs, n_hidden)
    nlp = neglogp(act, mu, logstd)
    entropy = (logstd + HALF_LOG_2PIE).sum(-1).mean()
    v_clip = old_v + torch.clamp(v - old_v, -cr, cr)
    vf_loss = 0.5 * torch.maximum((v - ret) ** 2, (v_clip - ret) ** 2).mean()
    ratio = torch.exp(old_nlp - nlp)
    pg_loss = torch.maximum(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - cr, 1.0 + cr)).mean()
    approxkl = 0.5 * ((nlp - old_nlp) ** 2).mean()
    clipfrac = ((ratio - 1.0).abs() > cr).to(obs.dtype).mean()
    loss = pg_loss - entropy * ent_coef + vf_loss * vf_coef
    return loss, (pg_loss, vf_loss, entropy, approxkl, clipfrac)


def loss_and_grads(named, n_hidden, obs, act, adv, ret, old_nlp, old_v, cr, ent_coef, vf_coef):
    p = as_params(named)
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float64)
    loss, parts = losses(p, n_hidden, t(obs), t(act), t(adv), t(ret), t(old_nlp), t(old_v), cr, ent_coef, vf_coef)
    loss.backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in p.items()}
    return np.array([float(x.detach()) for x in parts]), grads


def clip_and_adam(named, grads, m, v, pw, lr, max_norm, b1, b2, eps):
    """TF clip_by_global_norm + ApplyAdam in float64."""
    norm = np.sqrt(sum(float((g ** 2).sum()) for g in grads.values()))
    scale = max_norm * min(1.0 / norm, 1.0 / max_norm)
    alpha = lr * np.sqrt(1.0 - pw[1]) / (1.0 - pw[0])
    out_p, out_m, out_v = {}, {}, {}
    for k in named:
        g = grads[k] * scale
        out_m[k] = m[k] + (g - m[k]) * (1.0 - b1)
        out_v[k] = v[k] + (g * g - v[k]) * (1.0 - b2)
        out_p[k] = named[k] - (out_m[k] * alpha) / (np.sqrt(out_v[k]) + eps)
    return out_p, out_m, out_v, (pw[0] * b1, pw[1] * b2), norm
