import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

# PyTorch ships its own ROCm runtime and librccl.so.1; whichever copy a process loads first serves everybody in it.
# The pipe tests create an RCCL communicator through the C ABI: torch has to be there before librama_hip.so touches
# the GPU, whatever subset of the files is run (a whole-suite run got this from test_pipeline_gloo's import already).
try:
    import torch  # noqa: F401
except Exception:      # the suite's CPU part does not need it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return REPO / "tests" / "golden"
