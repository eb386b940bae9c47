"""Builds the in-tree native libraries for gfx950 (MI355X).  hipcc cross-compiles without a GPU.

    python -m ppo_cpp_amd.build            # libppo_hip.so (+ host library when present)
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
HIP_SO = os.path.join(PKG, "libppo_hip.so")
HOST_SO = os.path.join(PKG, "libppo_host.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return out


def build_hip(force=False, verbose=False):
    csrc = os.path.join(PKG, "csrc")
    deps = _sources(csrc, (".hip", ".hpp", ".h")) + [os.path.join(ROOT, "include", "ppo_hip.h")]
    if force or _newer(HIP_SO, deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
               "-o", HIP_SO, os.path.join(csrc, "ppo_hip.hip"), "-ldl"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_SO


def build_host(force=False, verbose=False):
    host = os.path.join(PKG, "host")
    src = os.path.join(host, "ppo_host.cpp")
    if not os.path.exists(src):
        return None
    deps = _sources(host, (".cpp", ".hpp", ".h")) + [os.path.join(ROOT, "include", "ppo_hip.h")]
    if force or _newer(HOST_SO, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-DPPO_MAT_NO_BOUNDS", "-I", os.path.join(ROOT, "include"),
               "-I", host, "-o", HOST_SO, src, "-L", PKG, "-lppo_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_SO


def build_driver(force=False, verbose=False):
    """ppo_cpp_hip: command-line driver with the reference's training flags (mock environments)."""
    host = os.path.join(PKG, "host")
    src = os.path.join(host, "main.cpp")
    exe = os.path.join(PKG, "ppo_cpp_hip")
    if not os.path.exists(src):
        return None
    deps = _sources(host, (".cpp", ".hpp", ".h")) + [os.path.join(ROOT, "include", "ppo_hip.h")]
    if force or _newer(exe, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-DPPO_MAT_NO_BOUNDS", "-I", os.path.join(ROOT, "include"), "-I", host, "-o", exe, src,
               "-L", PKG, "-lppo_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return exe


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose), build_driver(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
