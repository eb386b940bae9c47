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
