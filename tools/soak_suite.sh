#!/bin/bash
# the iteration-heavy cases that `pytest -m gpu` skips (tests/conftest.py: marker `slow`, option --runslow): the full-length determinism soaks and the largest row counts
cd "$(dirname "$0")/.." && exec python3 -m pytest tests -m "gpu and slow" --runslow -q -rxX "$@"
