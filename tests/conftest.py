import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


ROBOT_DIR = os.path.join(ROOT, "curobo_amd", "content", "robot")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle():
    from oracle import load_oracle

    return load_oracle()


def load_model(name):
    from curobo_amd.robot import RobotModel

    return RobotModel.load_npz(os.path.join(ROBOT_DIR, f"{name}.npz"))


@pytest.fixture(scope="session")
def franka():
    return load_model("franka")


@pytest.fixture(scope="session")
def ur10e():
    return load_model("ur10e")


@pytest.fixture(scope="session")
def g1():
    return load_model("unitree_g1")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def sample_q(model, n, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    lo, hi = model.joint_limits_position
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo) * scale
    return (mid + half * rng.uniform(-1, 1, size=(n, model.num_dof))).astype(np.float32)
