import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    arrays = np.load(os.path.join(GOLDEN_DIR, "crnn_golden.npz"))
    with open(os.path.join(GOLDEN_DIR, "crnn_golden.json"), encoding="utf-8") as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def synth_zh():
    from ocr_amd.crnn.synth import synth_state_dict
    return synth_state_dict(5530, seed=0)


@pytest.fixture(scope="session")
def synth_en():
    from ocr_amd.crnn.synth import synth_state_dict
    return synth_state_dict(65, seed=0)


@pytest.fixture(scope="session")
def oracle_zh(synth_zh):
    from oracle import oracle as O
    return O.Oracle(synth_zh, 5530)


@pytest.fixture(scope="session")
def oracle_en(synth_en):
    from oracle import oracle as O
    return O.Oracle(synth_en, 65)
