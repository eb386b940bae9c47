"""ppo_cpp_amd: MI355X-native (gfx950) PPO rollout-collect + minibatch-update hot path behind a C ABI.

The product is the native library ppo_cpp_amd/libppo_hip.so (hand-written HIP, see csrc/) declared in
include/ppo_hip.h; this package only binds it with ctypes.  There is no CPU fallback: constructing a
PPOHip without a usable gfx950 device raises.
"""
from .capi import PPOHip, PPOHipError, load_library  # noqa: F401
