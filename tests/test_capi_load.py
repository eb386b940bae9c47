"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol that
include/ppo_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ppo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppo_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_reference_call_sites():
    syms = declared_symbols()
    # one entry point per Session::Run call site of the reference (policies.hpp:37,53,68; ppo2.hpp:450; session_creator.hpp:54)
    for s in ("ppo_create", "ppo_step", "ppo_value", "ppo_act_deterministic", "ppo_train_step", "ppo_destroy", "ppo_last_error"):
        assert s in syms
    assert len(syms) >= 35


def test_library_builds_loads_and_exports_every_declared_symbol():
    import ppo_cpp_amd
    lib = ppo_cpp_amd.load_library()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.ppo_abi_version() == 1


def test_config_defaults_are_the_graph_baked_constants():
    import ppo_cpp_amd
    from ppo_cpp_amd.capi import PPOConfig
    from tests import helpers as H
    lib = ppo_cpp_amd.load_library()
    cfg = PPOConfig()
    hid = (ctypes.c_int32 * 2)(4, 5)
    lib.ppo_config_default(ctypes.byref(cfg), 18, 18, 2, hid)
    consts, _ = H.g45_consts()
    assert cfg.ent_coef == pytest.approx(consts["loss/mul_4/y"], rel=1e-7)
    assert cfg.vf_coef == consts["loss/mul_5/y"] and cfg.max_grad_norm == consts["loss/clip_by_global_norm/mul/x"]
    assert cfg.adam_beta1 == pytest.approx(consts["ppo2/_train/beta1"]) and cfg.adam_beta2 == pytest.approx(consts["ppo2/_train/beta2"])
    assert cfg.adam_eps == pytest.approx(consts["ppo2/_train/epsilon"])


def test_no_cpu_fallback_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ppo_cpp_amd
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="no CPU fallback"):
        ppo_cpp_amd.PPOHip(18, 18, [4, 5])


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under ppo_cpp_amd/ or include/ may reference it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "ppo_cpp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|ppo_oracle\.h|libppo_oracle|orc_[a-z_]+\s*\()", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
