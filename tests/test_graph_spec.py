"""Graph-spec importer: the C++ reader of the reference's MetaGraphDef text file recovers the network
shape, the graph-baked constants and the initial weights.  The 565 KB graph file itself is not copied into the repo; the
test runs where the reference tree is mounted (this container) and is skipped elsewhere; its expected values are the
committed fixtures (tests/golden/g45_init.npz) extracted independently by oracle/extract_fixtures.py."""
import ctypes as C
import glob

import numpy as np
import pytest

from ppo_cpp_amd import hostapi
from ppo_cpp_amd.capi import PPOConfig
from tests import helpers as H

GRAPHS = glob.glob("/root/reference/resources/ppo_cl/graphs/*.meta.txt")


@pytest.mark.skipif(not GRAPHS, reason="reference graph file not present on this machine")
def test_graph_spec_importer_matches_fixtures():
    lib = hostapi.load_host_library()
    cfg = PPOConfig(); pw = (C.c_float * 2)(); buf = np.zeros(4096, np.float32)
    consts, _ = H.g45_consts()
    init = H.g45_init()
    for name, ref in init.items():
        n = lib.ppo_host_graph_spec(GRAPHS[0].encode(), C.byref(cfg), pw, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
        assert n == ref.size, name
        np.testing.assert_array_equal(buf[:n], ref.reshape(-1))
    assert (cfg.obs_dim, cfg.act_dim, cfg.n_hidden, cfg.hidden[0], cfg.hidden[1]) == (18, 18, 2, 4, 5)
    assert cfg.ent_coef == np.float32(consts["loss/mul_4/y"]) and cfg.vf_coef == 0.5 and cfg.max_grad_norm == 0.5
    assert cfg.adam_beta1 == np.float32(0.9) and cfg.adam_beta2 == np.float32(0.999) and cfg.adam_eps == np.float32(1e-5)
    assert pw[0] == np.float32(0.9) and pw[1] == np.float32(0.999)          # beta powers start at beta (G:25426, 25579)


def test_graph_spec_reports_missing_file():
    lib = hostapi.load_host_library()
    cfg = PPOConfig(); pw = (C.c_float * 2)()
    assert lib.ppo_host_graph_spec(b"/nonexistent/graph.meta.txt", C.byref(cfg), pw, b"", None, 0) == -1


def write_meta_graph(path, init, consts):
    """A MetaGraphDef TEXT file in the layout the reference's generator emits (one `node { name / op / attr { key value
    { tensor {...} } } }` block per Const, variables initialised from `model/<v>/Initializer/...` constants, weights as
    C-escaped tensor_content), written from the COMMITTED fixtures -- so the importer and `ppo_cpp_hip -g` can be exercised
    on machines where the reference tree (and its 565 KB graph) is not mounted."""
    def const_node(name, arr=None, scalar=None):
        out = ['  node {', '    name: "%s"' % name, '    op: "Const"', '    attr {', '      key: "value"', '      value {', '        tensor {', '          dtype: DT_FLOAT',
               '          tensor_shape {']
        if arr is not None:
            for d in arr.shape:
                out += ['            dim {', '              size: %d' % d, '            }']
        out += ['          }']
        if arr is not None:
            raw = np.ascontiguousarray(arr, "<f4").tobytes()
            out += ['          tensor_content: "%s"' % "".join("\\%03o" % b for b in raw)]
        else:
            out += ['          float_val: %r' % float(scalar)]
        out += ['        }', '      }', '    }', '  }']
        return out
    lines = ['meta_info_def {', '  tensorflow_version: "1.14.0"', '}', 'graph_def {']
    for name, arr in init.items():
        lines += const_node("model/%s/Initializer/Const" % name, arr=np.asarray(arr, np.float32))
    for name in ("loss/mul_4/y", "loss/mul_5/y", "loss/clip_by_global_norm/mul/x", "ppo2/_train/beta1", "ppo2/_train/beta2", "ppo2/_train/epsilon"):
        lines += const_node(name, scalar=consts[name])
    lines += const_node("beta1_power/initial_value", scalar=consts["ppo2/_train/beta1"])
    lines += const_node("beta2_power/initial_value", scalar=consts["ppo2/_train/beta2"])
    lines += ['}']
    open(path, "w").write("\n".join(lines) + "\n")


def test_importer_reads_a_graph_written_from_the_fixtures(tmp_path):
    lib = hostapi.load_host_library()
    consts, _ = H.g45_consts(); init = H.g45_init()
    path = str(tmp_path / "g45.meta.txt"); write_meta_graph(path, init, consts)
    cfg = PPOConfig(); pw = (C.c_float * 2)(); buf = np.zeros(4096, np.float32)
    for name, ref in init.items():
        n = lib.ppo_host_graph_spec(path.encode(), C.byref(cfg), pw, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
        assert n == ref.size, name
        np.testing.assert_array_equal(buf[:n], np.asarray(ref, np.float32).reshape(-1))
    assert (cfg.obs_dim, cfg.act_dim, cfg.n_hidden, cfg.hidden[0], cfg.hidden[1]) == (18, 18, 2, 4, 5)
    assert cfg.ent_coef == np.float32(consts["loss/mul_4/y"]) and pw[0] == np.float32(consts["ppo2/_train/beta1"])


@pytest.mark.gpu
def test_create_from_graph_and_driver_flag_on_the_device(tmp_path):
    """SURVEY 8f row 4 on the GPU: load_graph + init (session_creator.hpp:40-58) from a graph file -> the handle evaluates
    exactly like the oracle holding the committed initial weights (tests/golden/g45_init.npz, the values embedded in the
    reference's own graph); then `ppo_cpp_hip -g <graph>` trains from them (initial entropy = 18 * 1.4189385, G's logstd = 0)."""
    import os
    import subprocess
    from oracle import oracle as o
    lib = hostapi.load_host_library()
    consts, _ = H.g45_consts(); init = H.g45_init()
    path = str(tmp_path / "g45.meta.txt"); write_meta_graph(path, init, consts)
    orc = o.Oracle(18, 18, [4, 5]); orc.set_tensors(init)
    obs = np.random.RandomState(3).uniform(-1, 1, (33, 18)).astype(np.float32)
    act = np.zeros((33, 18), np.float32); val = np.zeros(33, np.float32); pw = (C.c_float * 2)()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.ppo_host_graph_eval(path.encode(), fp(obs), 33, fp(act), fp(val), pw) == 0
    mu, v = orc.forward(obs)
    np.testing.assert_allclose(act, mu, rtol=1e-4, atol=1e-6); np.testing.assert_allclose(val, v, rtol=1e-4, atol=1e-6)
    assert pw[0] == np.float32(consts["ppo2/_train/beta1"]) and pw[1] == np.float32(consts["ppo2/_train/beta2"])
    from ppo_cpp_amd import build as b
    exe = b.build_driver() if os.path.exists("/opt/rocm/bin/hipcc") else os.path.join(os.path.dirname(hostapi.__file__), "ppo_cpp_hip")
    out = subprocess.run([exe, "-g", path, "--steps", "1024", "--batch_steps", "256", "--threads", "2", "--epochs", "2", "--minibatches", "4", "--lr", "3e-4",
                          "--cr", "0.2", "--seeded"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = [[float(x) for x in l.split(",")[:6]] for l in out.stdout.splitlines() if l.count(",") == 6]
    assert len(rows) == 2 and np.isfinite(rows).all()
    assert rows[0][3] == pytest.approx(18 * 1.4189385, rel=2e-3)
