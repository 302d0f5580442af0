import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHTS_FP32 = os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_fp32.npz")
WEIGHTS_TCNN = os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_tcnn.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # before any test initialises the HIP runtime: 8 hardware queues for the frame pipelines' streams
    from bnv_fusion_amd import configure_runtime
    configure_runtime()


def pytest_collection_modifyitems(config, items):
    """A hung GPU call or rendezvous must fail ONE test with a stack dump, not stall the whole run: every test gets a
    generous per-test limit when pytest-timeout is installed (thread method: it also fires inside a blocked C call)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
