"""Checkpoint I/O in the reference's on-disk format: the C++ bundle reader parses the reference's own
trained checkpoint (fixture copy of resources/ppo_cl/*.pkl.71.{index,data}) and the writer reproduces both files byte
for byte; tensors agree with the values the test-infrastructure reader (oracle/extract_fixtures.py) extracted."""
import ctypes as C
import os

import numpy as np

from ppo_cpp_amd import hostapi
from tests import helpers as H


def test_bundle_roundtrip_is_byte_exact(tmp_path):
    lib = hostapi.load_host_library()
    src = os.path.join(H.GOLDEN, "ckpt71")
    dst = str(tmp_path / "copy")
    assert lib.ppo_host_bundle_roundtrip(src.encode(), dst.encode()) == 15
    for ext in (".index", ".data-00000-of-00001"):
        assert open(src + ext, "rb").read() == open(dst + ext, "rb").read(), ext


def test_bundle_reader_matches_fixture_tensors():
    lib = hostapi.load_host_library()
    ck = H.ckpt71()
    buf = np.zeros(1024, np.float32)
    shape = (C.c_longlong * 4)()
    for name, ref in ck.items():
        n = lib.ppo_host_bundle_tensor(os.path.join(H.GOLDEN, "ckpt71").encode(), ("model/" + name).encode(),
                                       buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, shape)
        assert n == ref.size, name
        assert tuple(shape[i] for i in range(ref.ndim)) == ref.shape
        np.testing.assert_array_equal(buf[:n].reshape(ref.shape), ref)


def test_bundle_reader_rejects_corruption(tmp_path):
    lib = hostapi.load_host_library()
    src = os.path.join(H.GOLDEN, "ckpt71")
    for ext in (".index", ".data-00000-of-00001"):
        raw = bytearray(open(src + ext, "rb").read())
        raw[40] ^= 0x01
        p = str(tmp_path / "bad")
        for e2 in (".index", ".data-00000-of-00001"):
            open(p + e2, "wb").write(raw if e2 == ext else open(src + e2, "rb").read())
        assert lib.ppo_host_bundle_roundtrip(p.encode(), str(tmp_path / "out").encode()) == -1      # checksum mismatch


import pytest


@pytest.mark.gpu
def test_load_reference_checkpoint_eval_and_save(tmp_path):
    """PPO2::load of the reference's trained checkpoint (weights + JSON running statistics), deterministic action on the
    GPU equals the survey's sanity values (mu(0)[0..2] = -0.2594, -0.6368, -0.4799), PPO2::save reproduces the weight
    files byte for byte and a JSON with the same statistics."""
    import json
    import shutil
    lib = hostapi.load_host_library()
    src = str(tmp_path / "in")
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copyfile(os.path.join(H.GOLDEN, "ckpt71" + ext), src + ext)
    shutil.copyfile(os.path.join(H.GOLDEN, "ckpt71_stats.json"), src + ".json")
    mu = np.zeros(18, np.float32); cnt = C.c_double()
    dst = str(tmp_path / "out")
    assert lib.ppo_host_checkpoint_eval(src.encode(), dst.encode(), mu.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cnt)) == 0
    np.testing.assert_allclose(mu[:3], [-0.2594, -0.6368, -0.4799], atol=5e-5)
    assert cnt.value == pytest.approx(72001473.000001)
    for ext in (".index", ".data-00000-of-00001"):
        assert open(src + ext, "rb").read() == open(dst + ext, "rb").read(), ext
    a, b = json.load(open(src + ".json")), json.load(open(dst + ".json"))
    np.testing.assert_allclose(b["obs_rms"]["mean"], a["obs_rms"]["mean"], rtol=1e-7)
    np.testing.assert_allclose(b["ret_rms"]["var"], a["ret_rms"]["var"], rtol=1e-7)
    assert b["n_steps"] == 65536 and b["nminibatches"] == 32 and b["noptepochs"] == 10
