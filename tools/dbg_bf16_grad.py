"""Diagnostic: per-tensor gradient agreement (cosine, norm ratio) of the bf16 path with the fp32 oracle at [1024]^3, a few minibatch sizes."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as o
from tests import helpers as H
import ppo_cpp_amd
def cosine(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
hidden, O, A = (1024, 1024, 1024), 256, 64
for n in [int(x) for x in sys.argv[1:]] or [256, 1024]:
    orc = o.Oracle(O, A, list(hidden)); orc.init_orthogonal(3)
    orc.tensor("pi/logstd")[:] = np.random.RandomState(4).uniform(-1.0, 0.2, (1, A))
    mb = H.synth_minibatch(orc, n, seed=3)
    args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
    ref_losses, ref_grad = orc.loss_grad(*args, 0.16)
    res = {}
    for dt, nm in ((1, "bf16"), (0, "fp32")):
        g = ppo_cpp_amd.PPOHip(O, A, list(hidden), compute_dtype=dt); g.set_flat(orc.theta)
        g.train_step(3e-4, 0.16, *args); res[nm] = g.last_grad()[0].copy(); g.close()
    tot = np.linalg.norm(ref_grad)
    print("n =", n, "total norm", tot)
    for name, off, shape in orc.tensors:
        cnt = int(np.prod(shape)); rt = ref_grad[off:off + cnt]
        print("   %-12s rel norm %.3e   cos bf16 %.5f   cos fp32-hip %.7f   norm ratio bf16 %.4f" % (name, np.linalg.norm(rt) / tot, cosine(res["bf16"][off:off + cnt], rt),
              cosine(res["fp32"][off:off + cnt], rt), np.linalg.norm(res["bf16"][off:off + cnt]) / (np.linalg.norm(rt) + 1e-30)))
