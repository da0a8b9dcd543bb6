import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_03 = os.path.join(GOLDEN, "scenes", "03_volume", "volume.json")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vpt():
    import vpt_loader
    return vpt_loader.load()


@pytest.fixture(scope="session")
def scene03(vpt):
    return vpt.HostScene(SCENE_03)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def dev03(vpt, scene03):
    if vpt.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the HIP path has no CPU fallback)")
    return vpt.DeviceScene(scene03, 0)
