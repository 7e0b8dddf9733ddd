import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _class_switches_restored():
    """The product's class-level switches (layers.ConvBNAct.pointwise / conv3x3_forward / stem, FusedBNAct.enabled) are process
    state: a test that flips one and restores the wrong value changes what every later test of the run exercises (round 5 found
    tests/test_gpu_conv3x3_wgrad.py leaving conv3x3_forward = False behind).  Saved before and checked + restored after each test."""
    if "heltondetection_amd.layers" not in sys.modules:   # nothing to guard until a test imported the layers
        yield
        if "heltondetection_amd.layers" not in sys.modules:
            return
        from heltondetection_amd import layers
        defaults = {"pointwise": True, "conv3x3_forward": True, "stem": True}
        left = {k: getattr(layers.ConvBNAct, k) for k in defaults if getattr(layers.ConvBNAct, k) != defaults[k]}
        for k, v in defaults.items():
            setattr(layers.ConvBNAct, k, v)
        layers.FusedBNAct.enabled = True
        assert not left, f"test left ConvBNAct switches changed: {left}"
        return
    from heltondetection_amd import layers
    before = {k: getattr(layers.ConvBNAct, k) for k in ("pointwise", "conv3x3_forward", "stem")}
    before_bn = layers.FusedBNAct.enabled
    yield
    left = {k: getattr(layers.ConvBNAct, k) for k in before if getattr(layers.ConvBNAct, k) != before[k]}
    if layers.FusedBNAct.enabled != before_bn:
        left["FusedBNAct.enabled"] = layers.FusedBNAct.enabled
    for k, v in before.items():
        setattr(layers.ConvBNAct, k, v)
    layers.FusedBNAct.enabled = before_bn
    assert not left, f"test left class switches changed: {left}"
