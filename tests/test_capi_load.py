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
    assert lib.ppo_abi_version() == 3


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


def _build_c_program(tmp_path):
    import subprocess
    import ppo_cpp_amd
    ppo_cpp_amd.load_library()                                   # makes sure the .so exists
    src = tmp_path / "use_abi.c"
    src.write_text(C_PROGRAM)
    exe = tmp_path / "use_abi"
    libdir = os.path.join(ROOT, "ppo_cpp_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lppo_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


C_PROGRAM = r"""
#include <stdio.h>
#include <math.h>
#include "ppo_hip.h"
int main(void) {
    ppo_config cfg; int32_t hidden[2] = {256, 256}; ppo_handle* h = 0;
    float obs[18] = {0}, act[18], det[18], val[1], nlp[1], losses[5];
    int i;
    ppo_config_default(&cfg, 18, 18, 2, hidden);
    if (ppo_create(&cfg, &h) != 0) { printf("%s\n", ppo_last_error(0)); return 2; }
    if (ppo_init_orthogonal(h, 0) != 0) return 3;
    if (ppo_step(h, obs, 1, 0, act, val, nlp) != 0) { printf("%s\n", ppo_last_error(h)); return 4; }
    if (ppo_act_deterministic(h, obs, 1, det) != 0) return 5;
    for (i = 0; i < 18; ++i) if (!isfinite(act[i]) || !isfinite(det[i])) return 6;
    if (!isfinite(val[0]) || !(nlp[0] > 0.0f)) return 7;
    if (ppo_step(h, obs, 0, 0, act, val, nlp) == 0) return 8;         /* empty batch: refused, message available */
    if (ppo_last_error(h)[0] == 0) return 9;
    (void)ppo_train_step; (void)ppo_update; (void)ppo_gae; (void)ppo_norm_obs; (void)ppo_dist_init; (void)losses;
    ppo_destroy(h);
    printf("c abi ok\n");
    return 0;
}
"""


@pytest.mark.gpu
def test_c_program_runs_on_the_device(tmp_path):
    """The same C program, executed: create / init / step / deterministic action / error path / destroy from plain C."""
    import subprocess
    exe = _build_c_program(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "c abi ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/ppo_hip.h must be consumable from C (no C++-isms, no torch types): a C translation unit that calls the
    entry points compiles with gcc -std=c99 -pedantic -Werror and links against libppo_hip.so (link check only, no GPU)."""
    assert _build_c_program(tmp_path).exists()
