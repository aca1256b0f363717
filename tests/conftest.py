import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that waits for ever (a launch polling a counter that never moves, a faulted queue) must fail, not hold the box
    until its limit: every test gets a wall-clock bound through pytest-timeout when the plugin is present (it is in this image)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300 if item.get_closest_marker("gpu") else 600, method="thread"))   # "thread": also ends a process stuck inside a HIP call


@pytest.fixture(scope="session")
def weights():
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.weights import synthetic_weights
    return synthetic_weights(hp, seed=1234, perturb=True)
