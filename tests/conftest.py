import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name), "r") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def sample_robot():
    from gear_amd import urdf
    return urdf.robot_from_dict(load_golden("sample_robot.json"))


@pytest.fixture(scope="session")
def obstacles_robot():
    from gear_amd import urdf
    return urdf.robot_from_dict(load_golden("obstacles.json"))


@pytest.fixture(scope="session")
def collision_check_golden():
    return load_golden("collision_check.json")
