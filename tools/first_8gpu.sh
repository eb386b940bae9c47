#!/bin/bash
# ONE command for the first multi-device run: see tools/first_8gpu.py.  The parent stays off the GPU; every step is a fresh child process.
cd "$(dirname "$0")/.." && export HSA_ENABLE_IPC_MODE_LEGACY=0 && exec python3 tools/first_8gpu.py "$@"
