import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(HERE, "golden", "goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fixture_images():
    from cases import images
    return images()
