import os
import sys

import numpy as np
import pytest

from conftest import load_pkg


def test_spec_counts():
    s = load_pkg("synth")
    for c in (32, 48):
        spec = s.hrnet_state_spec(c, 17)
        assert len(spec) == 1754
        convs = [x for x in spec if x[2] == "conv"]
        assert len(convs) == 293
    n32 = sum(int(np.prod(sh)) for _, sh, k in s.hrnet_state_spec(32, 17) if k != "bn_count")
    n48 = sum(int(np.prod(sh)) for _, sh, k in s.hrnet_state_spec(48, 17) if k != "bn_count")
    # 28.54 M / 63.60 M parameters + BN running stats
    assert 28.4e6 < n32 < 28.8e6 and 63.4e6 < n48 < 63.9e6


def test_deterministic():
    s = load_pkg("synth")
    a, b = s.synth_state_dict(32, 17, 5), s.synth_state_dict(32, 17, 5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["conv1.weight"], s.synth_state_dict(32, 17, 6)["conv1.weight"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/models_"), reason="reference only exists in the build container")
def test_spec_matches_reference_state_dict():
    sys.path.insert(0, "/root/reference")
    from models_.hrnet import HRNet

    s = load_pkg("synth")
    for c in (32, 48):
        ref = HRNet(c, 17).state_dict()
        spec = s.hrnet_state_spec(c, 17)
        assert [k for k, _, _ in spec] == list(ref.keys())
        assert all(tuple(ref[k].shape) == sh for k, sh, _ in spec)
