import sys, os
sys.path.insert(0, os.getcwd())
import pytest
from tests import test_other_shapes as t
mp = pytest.MonkeyPatch()
t._il_probe(lambda l: print(l, flush=True), mp)
