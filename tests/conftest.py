import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library ignores every HRN_* switch unless the process opts in (kernels.h: hrn_env); the bit-identity tests flip them.
os.environ["HRN_DEBUG_ENV"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg(sub=None):
    name = "simple-hrnet_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def synth():
    return load_pkg("synth")


_SD_CACHE = {}


def state_dict_np(c, seed=0):
    key = (c, seed)
    if key not in _SD_CACHE:
        _SD_CACHE[key] = load_pkg("synth").synth_state_dict(c, 17, seed)
    return _SD_CACHE[key]
